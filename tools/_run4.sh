set -u
mkdir -p gpurun_out/r5b
timeout 2400 python -m pytest tests/test_step_gpu.py tests/test_dp_gpu.py -x -q -k "ordering or argument_checks or rccl or overflow or config4 or chunked or train_loop_follows or two_ranks or layerwise" > gpurun_out/r5b/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r5b/pytest.log
for c in 3 4; do timeout 600 python bench.py --config $c --steps 20 --warmup 5 --no-pmc > gpurun_out/r5b/bench_c$c.json 2> gpurun_out/r5b/bench_c$c.err; echo "config $c rc=$?"; python -c "
import json;d=json.loads(open('gpurun_out/r5b/bench_c$c.json').read().strip().splitlines()[-1]);print(d['metric'],d['value'],d['ms_per_step'],d['dtype'],d['config']['workload']);print(d.get('parity'));print(d['cpu_baseline']['value'])"; done
