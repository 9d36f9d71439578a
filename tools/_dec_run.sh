R=/root/repo
mkdir -p $R/gpurun_out/dec1
timeout 900 python -m pytest $R/tests/test_decode_step_gpu.py -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/decprof -o dec -- python $R/tools/decode_phase_profile.py run > $R/gpurun_out/dec1/run.txt 2>&1
python $R/tools/decode_phase_profile.py report /tmp/decprof > $R/gpurun_out/dec1/phases.txt 2>&1
cat $R/gpurun_out/dec1/phases.txt
cd $R && timeout 600 python tools/bench_decode.py --modes graph-phases --gen 1024 2>&1 | tail -1
