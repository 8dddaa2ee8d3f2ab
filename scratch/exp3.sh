for cap in 1024 2048 4096 8192 1000000; do for srt in "" 1; do
  echo "cap=$cap sorted=$srt d=128: $(env SORTED=$srt D=128 GD_SPMM_GRID_CAP=$cap python scratch/spmm_micro.py 2>&1 | tail -1)  d=64: $(env SORTED=$srt D=64 GD_SPMM_GRID_CAP=$cap python scratch/spmm_micro.py 2>&1 | tail -1)"
done; done
