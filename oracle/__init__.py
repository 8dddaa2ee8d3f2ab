"""CPU oracle for the GNNDelete hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; nothing under ``gnndelete_amd/`` does
(``tests/test_layout.py`` enforces it).

Two halves, pinned differently (see DESIGN.md "Oracle"):

* ``oracle.gnndelete_ref``  - restatement of code the reference owns
  (DeletionLayer, *Delete wiring, loss zoo, update rules, eval).  PINNED against
  the reference itself: ``tests/golden/make_golden.py`` imports
  ``/root/reference/framework`` under third-party stubs, runs the reference's
  own classes/loops, and commits the outputs as ``tests/golden/*.npz``.
* ``oracle.pyg_semantics``  - restatement of the un-vendored torch_geometric
  arithmetic the reference calls (GCNConv/GATConv/GINConv/RGCNConv, k_hop_subgraph,
  to_undirected).  torch_geometric is absent from /root/reference and from this
  image and the reference has no tests: **parity unpinned** at that boundary.
  It is pinned instead by closed-form dense known-answer tests
  (``tests/test_oracle_kat.py``).
"""
