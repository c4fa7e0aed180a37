R=$GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -x -k "small_image or chunk_major or epilogue_stat or accumulators" 2>&1 | tail -3
python -m pytest tests/test_round2_gpu.py -q -x -k "resblock" 2>&1 | tail -3
for r in 1 2; do
DIFFSEP_NO_SMALL=1 python bench.py --no-extra-modes --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('base ', d['value'], d['one_batch_alone_ms'])"
python bench.py --no-extra-modes --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('small', d['value'], d['one_batch_alone_ms'])"
done
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/r2s
rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/r2s/prof -- python $R/bench.py --no-extra-modes --in-flight 1 --steps 2 --warmup 1 > $R/gpurun_out/r2s/bench.log 2>&1
DB=$(find $R/gpurun_out/r2s/prof -name "*.db" | head -1)
python $R/tools/rocpd_by_shape.py $DB $R/gpurun_out/r2s/by_shape.md 60 | grep -i "small" 
rm -rf $R/gpurun_out/r2s/prof
