#!/usr/bin/env python3
"""Node-feature unlearning CLI (reference: delete_node_feature.py, which differs from delete_node.py by
five lines: checkpoint root `checkpoint_node_feature`, the Df nodes' feature rows are zeroed instead of
the nodes being dropped from the retained-node mask)."""
from delete_node import main

if __name__ == '__main__':
    main(feature_only=True)
