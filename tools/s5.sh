#!/bin/bash
# precision A/B of the round-5 kernel variants (same box): which change costs agreement with the fp32 engine?
cd /root/repo; mkdir -p gpurun_out diffusion-separation_amd/abl
CS=diffusion-separation_amd/csrc
HC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -DDS_HALF_F16 -I$PWD/$CS"
mk() {  # name, rw source, rw flags, norm flags
  $HC $3 -mllvm -pragma-unroll-threshold=1000000 -c $2 -o /tmp/v_rw.o && $HC $4 -c $CS/norm.hip -o /tmp/v_norm.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o diffusion-separation_amd/abl/lib_$1.so /tmp/v_rw.o /tmp/v_norm.o $(ls $CS/build_f16/*.o | grep -Ev '/(conv3x3_rw|norm)\.o$')
}
cp tools/ab_src/conv3x3_rw_r04.hip /tmp/rw_r04.hip
mk r04 /tmp/rw_r04.hip "" "-DDS_NO_FIR_TILED"
mk new_all $CS/conv3x3_rw.hip "" ""
mk new_rw_oldfir $CS/conv3x3_rw.hip "" "-DDS_NO_FIR_TILED"
mk new_rw_f32act_oldfir $CS/conv3x3_rw.hip "-DRW_ACT_F32" "-DDS_NO_FIR_TILED"
mk r04rw_newfir /tmp/rw_r04.hip "" ""
for v in r04 new_all new_rw_oldfir new_rw_f32act_oldfir r04rw_newfir; do
  DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_$v.so python tools/precision_probe.py 64 16 2>&1 | grep -v amdgpu
  DIFFSEP_LIB_F16=$PWD/diffusion-separation_amd/abl/lib_$v.so python tools/precision_probe.py 128 4 2>&1 | grep -v amdgpu
done > gpurun_out/precision_s5.txt 2>&1
cat gpurun_out/precision_s5.txt
