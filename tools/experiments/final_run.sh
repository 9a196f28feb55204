#!/bin/bash
# The round's records, taken ONCE on the final sources: GPU test suite, default bench line, rocprofv3 kernel stats + one step's
# timeline, PMC passes, the other configurations (Alimama / gate / bilinear / one rank through RCCL), the N-rank compute emulation,
# phase stamps.  Output under gpurun_out/r05_final; copy what is to be kept into profiles/.
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
o=gpurun_out/r05_final; mkdir -p $o
python -m pytest tests -m gpu -q > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $o/ks -o p -- python3 bench.py --train-only --steps 25 --warmup 5 --no-phase-timing > $o/ks.log 2>&1
python tools/kernel_stats.py $(find $o/ks -name "*kernel_stats.csv" | head -1) 40 > $o/kernel_stats.md
cp $(find $o/ks -name "*kernel_stats.csv" | head -1) $o/kernel_stats.csv
python tools/step_timeline.py $(find $o/ks -name "*kernel_trace.csv" | head -1) 12 > $o/timeline.txt
rm -rf $o/ks
bash tools/pmc_passes.sh > $o/pmc.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc aliccp > $o/pmc_summary.json 2> $o/pmc_summary.err
rm -rf gpurun_out/pmc
# the default bench line AFTER the counter passes: bench.py takes roofline.traffic from profiles/r06_pmc_summary.json and refuses a
# summary of other kernel sources (on the GPU box this copy only lives for the call; copy it into profiles/ here as well)
cp $o/pmc_summary.json profiles/r06_pmc_summary.json
python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python bench.py --config alimama --train-only > $o/bench_alimama.json 2>/dev/null
python bench.py --flag sota-gate --train-only > $o/bench_gate.json 2>/dev/null
python bench.py --flag sota-bilinear --train-only > $o/bench_bilinear.json 2>/dev/null
SATRANS_FORCE_EXCHANGE=1 python bench.py --train-only > $o/bench_owner.json 2>/dev/null
python tools/fake_world.py 1 2 4 8 > $o/fake_world_owner.txt 2>/dev/null
python tools/stamps.py > $o/stamps.txt 2>&1
head -c 600 $o/bench_default.json
