#!/bin/bash
# usage (GPU box): tools/accuracy_evidence.sh [1|2a|2b|3] -- the accuracy record behind bench.py's `tolerance_met` (gpurun_out/r06_accuracy_teacher*.txt;
# the parts are joined into profiles/r06_accuracy_teacher.txt by tools/accuracy_join.py afterwards): to be re-run whenever a file of TEACHER_CSRC /
# TEACHER_HOST changes (the record's first line carries their hash).  One part per gpurun call (a call is limited to 20 minutes).
#   part 1:  tests/test_precision_gpu.py with all seven weight / batch seeds at 224^2 and 448^2, four at 640^2, PAR, four batches of b = 16
#   part 2a: 40 held-out draws at 448^2 (seeds 100-139: the set round 5's default was chosen on), four modes
#   part 2b: 32 more (seeds 200-231: drawn after that choice) + 24 new in round 6 (300-323: drawn after the criterion was pre-registered)
#   part 3:  32 draws at 224^2, 12 at 640^2, three modes
#   part 4:  40 draws at 448^2 with seeds 400-439, drawn AFTER the default mode was chosen (fp16x3, fp16c8-x2, bf16x3), + the four b = 16 batches for bf16x3
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repository root on the GPU box)}" || exit 1
W=${1:?part: 1 | 2a | 2b | 3 | 4}
R=gpurun_out/r06_accuracy_teacher.txt
rm -f $R gpurun_out/r06_accuracy_teacher_par.txt
run() { "$@" > gpurun_out/evidence_part$W.log 2>&1; tail -1 gpurun_out/evidence_part$W.log; }          # (a failing draw is on record: the run goes on)
M4=fp16x3,fp16c8-x2,fp16c8,bf16x3
M3=fp16x3,fp16c8-x2,fp16c8
case $W in
1)  COSA_ACCURACY_ALL_SEEDS=1 COSA_ACCURACY_B16_SEEDS=7,8,9,10 COSA_ACCURACY_B16_MODES=fp16x3,fp16c8-x2 COSA_ACCURACY_SWEEP_SEEDS=1 run python -m pytest tests/test_precision_gpu.py -q -m gpu
    sed -i "1a # part 1: tests/test_precision_gpu.py with COSA_ACCURACY_ALL_SEEDS=1 COSA_ACCURACY_B16_SEEDS=7,8,9,10: seven seeds at 224^2 / 448^2, four at 640^2, one sweep draw, four b = 16 batches" $R ;;
2a) echo "# part 2a: COSA_ACCURACY_SWEEP_SEEDS=40 COSA_ACCURACY_SWEEP_BASE=100, modes $M4 (pytest -k sweep)" > $R.hdr
    COSA_ACCURACY_SWEEP_MODES=$M4 COSA_ACCURACY_SWEEP_SEEDS=40 COSA_ACCURACY_SWEEP_BASE=100 run python -m pytest tests/test_precision_gpu.py -q -m gpu -k "sweep or pooled" ;;
2b) echo "# part 2b: COSA_ACCURACY_SWEEP_SEEDS=32 COSA_ACCURACY_SWEEP_BASE=200 and COSA_ACCURACY_SWEEP_SEEDS=24 COSA_ACCURACY_SWEEP_BASE=300, modes $M4 (pytest -k sweep)" > $R.hdr
    COSA_ACCURACY_SWEEP_MODES=$M4 COSA_ACCURACY_SWEEP_SEEDS=32 COSA_ACCURACY_SWEEP_BASE=200 run python -m pytest tests/test_precision_gpu.py -q -m gpu -k "sweep or pooled"
    COSA_ACCURACY_SWEEP_MODES=$M4 COSA_ACCURACY_SWEEP_SEEDS=24 COSA_ACCURACY_SWEEP_BASE=300 run python -m pytest tests/test_precision_gpu.py -q -m gpu -k "sweep or pooled" ;;
3)  echo "# part 3: COSA_ACCURACY_SWEEP_S=224 COSA_ACCURACY_SWEEP_SEEDS=32 and COSA_ACCURACY_SWEEP_S=640 COSA_ACCURACY_SWEEP_SEEDS=12 (seeds 100 + i), modes $M3" > $R.hdr
    COSA_ACCURACY_SWEEP_MODES=$M3 COSA_ACCURACY_SWEEP_S=224 COSA_ACCURACY_SWEEP_SEEDS=32 run python -m pytest tests/test_precision_gpu.py -q -m gpu -k "sweep or pooled"
    COSA_ACCURACY_SWEEP_MODES=$M3 COSA_ACCURACY_SWEEP_S=640 COSA_ACCURACY_SWEEP_SEEDS=12 run python -m pytest tests/test_precision_gpu.py -q -m gpu -k "sweep or pooled" ;;
4)  echo "# part 4: COSA_ACCURACY_SWEEP_SEEDS=40 COSA_ACCURACY_SWEEP_BASE=400 (drawn after the default was chosen), modes fp16x3,fp16c8-x2,bf16x3; COSA_ACCURACY_B16_SEEDS=7,8,9,10 for bf16x3" > $R.hdr
    COSA_ACCURACY_SWEEP_MODES=fp16x3,fp16c8-x2,bf16x3 COSA_ACCURACY_SWEEP_SEEDS=40 COSA_ACCURACY_SWEEP_BASE=400 run python -m pytest tests/test_precision_gpu.py -q -m gpu -k "sweep or pooled"
    COSA_ACCURACY_B16_SEEDS=7,8,9,10 COSA_ACCURACY_B16_MODES=bf16x3 run python -m pytest tests/test_precision_gpu.py -q -m gpu -k "b16 or pooled" ;;
esac
[ -f $R.hdr ] && { sed -i "1r $R.hdr" $R; rm -f $R.hdr; }
cp $R gpurun_out/r06_accuracy_teacher_part$W.txt
python tools/accuracy_worst.py $R
