#!/bin/bash
# The round's records, taken ONCE on the final sources: smoke, GPU test suite, rocprofv3 kernel stats + one step's timeline, PMC passes
# (AliCCP: four passes; configs[4]: FETCH_SIZE / WRITE_SIZE), the default bench line AFTER the counter passes (it takes
# roofline.traffic from profiles/r06_*pmc_summary.json and refuses a summary of other kernel sources), configs[4] kernel stats,
# the other single lines (bilinear, one rank through RCCL), the N-rank compute emulation.  Output under gpurun_out/r06_final; copy
# what is to be kept into profiles/.
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r06_final; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $o/smoke.log
python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $o/ks -o p -- python3 bench.py --train-only --steps 25 --warmup 5 --no-phase-timing > $o/ks.log 2>&1
python tools/kernel_stats.py $(find $o/ks -name "*kernel_stats.csv" | head -1) 40 > $o/kernel_stats.md
cp $(find $o/ks -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
python tools/step_timeline.py $(find $o/ks -name "*kernel_trace.csv" | head -1) 12 > $o/timeline.txt
rm -rf $o/ks
bash tools/pmc_passes.sh > $o/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc aliccp > $o/pmc_summary.json 2> $o/pmc_summary.err
rm -rf gpurun_out/pmc
cp $o/pmc_summary.json profiles/r06_pmc_summary.json
PMC_OUT=gpurun_out/pmc_alimama PMC_BENCH_ARGS="--config alimama" PMC_ONLY_TRAFFIC=1 bash tools/pmc_passes.sh > $o/pmc_alimama.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_alimama alimama > $o/alimama_pmc_summary.json 2> $o/alimama_pmc_summary.err
rm -rf gpurun_out/pmc_alimama
cp $o/alimama_pmc_summary.json profiles/r06_alimama_pmc_summary.json
PMC_OUT=gpurun_out/pmc_gate PMC_BENCH_ARGS="--flag sota-gate" PMC_ONLY_TRAFFIC=1 bash tools/pmc_passes.sh > $o/pmc_gate.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_gate aliccp:sota-gate > $o/gate_pmc_summary.json 2> $o/gate_pmc_summary.err
rm -rf gpurun_out/pmc_gate
cp $o/gate_pmc_summary.json profiles/r06_gate_pmc_summary.json
bash tools/pmc_c5.sh > $o/pmc_c5.log 2>&1
python tools/pmc_c5_summary.py gpurun_out/pmc_c5 6 > $o/c5_pmc_summary.json 2> $o/c5_pmc_summary.err
rm -rf gpurun_out/pmc_c5
cp $o/c5_pmc_summary.json profiles/r06_c5_pmc_summary.json
( time python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2> $o/bench_default.time; echo "bench rc=$?"; tail -3 $o/bench_default.time
rocprofv3 --kernel-trace --stats --output-format csv -d $o/ks5 -o p -- python3 bench.py --config c5 --train-only --steps 10 --warmup 2 --no-phase-timing > $o/ks5.log 2>&1
python tools/kernel_stats.py $(find $o/ks5 -name "*kernel_stats.csv" | head -1) 45 > $o/c5_kernel_stats.md
cp $(find $o/ks5 -name "*kernel_stats.csv" | head -1) $o/c5_kernel_stats.csv
rm -rf $o/ks5
python bench.py --flag sota-bilinear --train-only > $o/bench_bilinear.json 2>/dev/null
SATRANS_FORCE_EXCHANGE=1 python bench.py --train-only > $o/bench_owner.json 2>/dev/null
SATRANS_FORCE_EXCHANGE=1 SATRANS_OWNER_PREFETCH=1 python bench.py --train-only > $o/bench_owner_prefetch.json 2>/dev/null
SATRANS_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --train-only --steps 10 --warmup 3 > $o/bench_two_ranks_one_gpu_gloo.json 2> $o/bench_two_ranks.err; echo "two ranks rc=$?"
python tools/fake_world.py 1 2 4 8 > $o/fake_world_owner.txt 2>/dev/null
python tools/experiments/r06_fit_epoch.py > $o/fit_epoch.txt 2>/dev/null
head -c 700 $o/bench_default.json
