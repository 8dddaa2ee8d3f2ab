for srt in "" 1; do for v in GROUP PERSIST; do
  for d in 128 64; do
  echo "sorted=$srt variant=$v d=$d: $(env SORTED=$srt D=$d GD_SPMM_$v=1 python scratch/spmm_micro.py 2>&1 | tail -1)"
  done
done; done
