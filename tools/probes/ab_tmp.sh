python -m pytest tests/test_kernels_gpu.py -q -x -k "first_layer or conv3x3" 2>&1 | tail -3
python -m pytest tests/test_engine_gpu.py -q -x -k "golden or full_size_score or narrow" 2>&1 | tail -2
for r in 1 2; do python bench.py --no-extra-modes --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['one_batch_alone_ms'])"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2tl
rocprofv3 --kernel-trace --output-format rocpd -d $R/gpurun_out/r2tl/prof -- python $R/bench.py --no-extra-modes --in-flight 1 --steps 2 --warmup 1 > $R/gpurun_out/r2tl/bench.log 2>&1
DB=$(find $R/gpurun_out/r2tl/prof -name "*.db" | head -1)
python $R/tools/rocpd_by_shape.py $DB $R/gpurun_out/r2tl/by_shape.md 80 | grep "thin" | cut -c1-150
rm -rf $R/gpurun_out/r2tl/prof
