set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
# 1. driver-shaped bench line (all legs)
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
# 2. per-launch CSV of a 5-step headline run
CALD_PROFILE_DUMP=$GRAFT_REPO_ROOT/gpurun_out/final/launches_fp32.csv python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-pool --no-f16x3 --no-train --no-cfg4 > gpurun_out/final/bench5.json 2>/dev/null
# 3. whole-pool parity vs stored torch-CPU scores
python tools/parity_full_pool.py profiles/torch_cpu_reference_r2.npz gpurun_out/final/parity.json > gpurun_out/final/parity.log 2>&1
# 4. rocprof passes of the bench command
bash tools/profile_gpu.sh r6 > gpurun_out/final/profile.log 2>&1
# 5. training: kernel stats + event timeline + A/B of the pack plan
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final/train_prof -o t -- python $GRAFT_REPO_ROOT/tools/bench_train.py --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/final/train_prof.log 2>&1)
rm -f gpurun_out/final/train_prof/*kernel_trace.csv
python tools/train_event_timeline.py > gpurun_out/final/train_timeline.txt 2>&1
tools/ab_train.sh "CALD_TRAIN_PACK_PLAN=0" "CALD_TRAIN_PACK_PLAN=1" 3 > gpurun_out/final/train_ab.txt 2>&1
python tools/bench_train.py --steps 30 --warmup 6 > gpurun_out/final/bench_train.json 2> gpurun_out/final/bench_train.err
python tools/bench_train.py --model retinanet --steps 30 --warmup 6 > gpurun_out/final/bench_train_retinanet.json 2>/dev/null
tail -c 600 gpurun_out/final/bench.json
