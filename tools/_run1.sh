cd /root/repo
echo "== register-weight 128-cout kernel"; DIFFSEP_RW_RES=1 python tools/rw_bench.py 10 "128->128" 2>&1 | grep -v amdgpu
echo "== generic tile"; DIFFSEP_NO_RW128=1 python tools/rw_bench.py 10 "128->128" 2>&1 | grep -v amdgpu
python bench.py --no-cpu-baseline --no-extra-modes > gpurun_out/b_rw128.log 2>&1
python bench.py --no-cpu-baseline --no-extra-modes --nf 128 --in-flight 2 > gpurun_out/b_rw128_nf128.log 2>&1
DIFFSEP_NO_RW128=1 python bench.py --no-cpu-baseline --no-extra-modes --nf 128 --in-flight 2 > gpurun_out/b_norw128_nf128.log 2>&1
