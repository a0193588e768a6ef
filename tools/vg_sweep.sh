mkdir -p gpurun_out/r03g
python tools/debug_vgru.py > gpurun_out/r03g/debug_vgru.txt 2>&1
for cgs in 1 2 4; do
  echo "== DMP_VGRU_CGS=$cgs (NW auto)"; DMP_VGRU_CGS=$cgs python tools/time_vgru_group.py 4 300 2000 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r03g/sweep.txt 2>&1
for cgs in 2 4; do
  echo "== DMP_VGRU_CGS=$cgs DMP_VGRU_NW=2"; DMP_VGRU_NW=2 DMP_VGRU_CGS=$cgs python tools/time_vgru_group.py 4 300 2000 2>&1 | grep -v amdgpu.ids
done >> gpurun_out/r03g/sweep.txt 2>&1
tail -4 gpurun_out/r03g/debug_vgru.txt; cat gpurun_out/r03g/sweep.txt
