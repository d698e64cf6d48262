#!/bin/bash
# MFMA utilisation + HBM traffic of the bench forward in the headline mode (round 5: an ADAPTIVE mode -- the passes run 10 timed
# steps, i.e. the schedule's tiers in their proportions, stride 7 over the 50 (t, t_prev) pairs): three rocprofv3 --pmc passes
# (SQ counters; FETCH_SIZE; WRITE_SIZE -- counters only + --kernel-trace, as MI355X_MICROARCH.md prescribes) over
# `python bench.py --precision $PREC`.      IVID_COMMIT=$(git rev-parse --short HEAD) PREC=fp16sa3 bash scripts/r5/gpu_pmc.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
PREC=${PREC:-fp16sa3}
STEPS=${STEPS:-10}
D=gpurun_out/pmc_r5_$PREC
mkdir -p $D
export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  rm -rf $D/$name
  IVID_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $D/$name -o p -- \
    python bench.py --precision $PREC --steps $STEPS --warmup 1 --no-cpu-baseline --no-kernel-breakdown --no-parity-mode > $D/$name.log 2>&1
  echo "$name exit $?"
}
run SQ SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
python scripts/r4/pmc_mfma_summary.py $D $PREC | sed "s/--steps 1 --warmup 1/--steps $STEPS --warmup 1/" > $D/mfma.json && head -60 $D/mfma.json
python scripts/pmc_traffic.py $D $PREC | sed "s/--steps 1 --warmup 1/--steps $STEPS --warmup 1/" > $D/traffic.json && head -c 1500 $D/traffic.json
find $D -name "*.csv" -size +3M -delete
