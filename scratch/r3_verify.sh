#!/bin/bash
# final check of the round: whole GPU suite, smoke, the default bench line and configs[4]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=6 > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/steps.log
grep -q "tests rc=0" $O/steps.log || { tail -30 $O/tests.log; exit 1; }
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/steps.log
timeout -k 10 500 python3 bench.py > $O/bench_n100000.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/steps.log
timeout -k 10 500 python3 bench.py --total-images 250000 --steps 2 --no-cpu-baseline > $O/bench_n250000.json 2> $O/bench250.err; echo "bench250 rc=$?" | tee -a $O/steps.log
tail -4 $O/tests.log; tail -2 $O/smoke.log
