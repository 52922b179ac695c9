#!/bin/bash
# usage (GPU box): tools/accuracy_evidence.sh [all|1|23] -- the accuracy record behind bench.py's `tolerance_met` (gpurun_out/r05_accuracy_teacher.txt; copy it to
# profiles/ afterwards): to be re-run whenever a file of TEACHER_CSRC changes (the record's first line carries their hash).
#   part 1: tests/test_precision_gpu.py with all seven weight / batch seeds at 224^2 and 448^2, four at 640^2, PAR, the b = 16 batch
#   part 2: 72 held-out draws at 448^2 (seeds 100-139: the selection set; 200-231: drawn after the choice), four modes
#   part 3: 32 draws at 224^2, 12 at 640^2, three modes
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
R=gpurun_out/r05_accuracy_teacher.txt
W=${1:-all}          # all | 1 | 23 (a gpurun call is limited to 20 minutes: part 1 and parts 2-3 fit one call each)
if [ "$W" != 23 ]; then
rm -f $R gpurun_out/r05_accuracy_teacher_par.txt
COSA_ACCURACY_ALL_SEEDS=1 COSA_ACCURACY_B16=1 COSA_ACCURACY_SWEEP_SEEDS=3 python -m pytest tests/test_precision_gpu.py -q -m gpu -x > gpurun_out/evidence_part1.log 2>&1 || { tail -20 gpurun_out/evidence_part1.log; exit 1; }
tail -1 gpurun_out/evidence_part1.log
sed -i "1a # part 1: tests/test_precision_gpu.py with COSA_ACCURACY_ALL_SEEDS=1 COSA_ACCURACY_B16=1: seven seeds at 224^2 / 448^2, four at 640^2, three sweep draws, the b = 16 batch" $R
fi
[ "$W" = 1 ] && exit 0
[ "$W" = 23 ] && rm -f $R          # (gpurun_out/ does not travel to the box: parts 2-3 start a file of their own, joined to part 1 afterwards)
echo "# part 2: COSA_ACCURACY_SWEEP_SEEDS=40 COSA_ACCURACY_SWEEP_BASE=100 and COSA_ACCURACY_SWEEP_SEEDS=32 COSA_ACCURACY_SWEEP_BASE=200, modes fp16c8-x2, fp16c8, fp16c4-12m9, bf16x3 (pytest -k sweep)" >> $R
M4=fp16c8-x2,fp16c8,fp16c4-12m9,bf16x3
COSA_ACCURACY_SWEEP_MODES=$M4 COSA_ACCURACY_SWEEP_SEEDS=40 COSA_ACCURACY_SWEEP_BASE=100 python -m pytest tests/test_precision_gpu.py -q -m gpu -k sweep > gpurun_out/evidence_part2a.log 2>&1 || { tail -20 gpurun_out/evidence_part2a.log; exit 1; }
tail -1 gpurun_out/evidence_part2a.log
COSA_ACCURACY_SWEEP_MODES=$M4 COSA_ACCURACY_SWEEP_SEEDS=32 COSA_ACCURACY_SWEEP_BASE=200 python -m pytest tests/test_precision_gpu.py -q -m gpu -k sweep > gpurun_out/evidence_part2b.log 2>&1 || { tail -20 gpurun_out/evidence_part2b.log; exit 1; }
tail -1 gpurun_out/evidence_part2b.log
echo "# part 3: COSA_ACCURACY_SWEEP_S=224 COSA_ACCURACY_SWEEP_SEEDS=32 and COSA_ACCURACY_SWEEP_S=640 COSA_ACCURACY_SWEEP_SEEDS=12 (seeds 100 + i), modes fp16c8-x2, fp16c8, fp16c4-12m9" >> $R
M3=fp16c8-x2,fp16c8,fp16c4-12m9
COSA_ACCURACY_SWEEP_MODES=$M3 COSA_ACCURACY_SWEEP_S=224 COSA_ACCURACY_SWEEP_SEEDS=32 python -m pytest tests/test_precision_gpu.py -q -m gpu -k sweep > gpurun_out/evidence_part3a.log 2>&1 || { tail -20 gpurun_out/evidence_part3a.log; exit 1; }
tail -1 gpurun_out/evidence_part3a.log
COSA_ACCURACY_SWEEP_MODES=$M3 COSA_ACCURACY_SWEEP_S=640 COSA_ACCURACY_SWEEP_SEEDS=12 python -m pytest tests/test_precision_gpu.py -q -m gpu -k sweep > gpurun_out/evidence_part3b.log 2>&1 || { tail -20 gpurun_out/evidence_part3b.log; exit 1; }
tail -1 gpurun_out/evidence_part3b.log
python tools/accuracy_worst.py $R
