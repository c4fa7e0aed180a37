cd /root/repo
U="-mllvm -pragma-unroll-threshold=1000000"
echo "=== shipped"; RW_EXTRA="$U" bash tools/rw_timing.sh 2>&1 | grep -v amdgpu
echo "=== core only"; RW_EXTRA="$U -DRW_ABL_NOEPI -DRW_ABL_NOSTAGE" bash tools/rw_timing.sh 2>&1 | grep -v amdgpu
