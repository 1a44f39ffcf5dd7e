#!/bin/bash
# Same-lease A/B of the ramp group's write-back (round 6, second session): every C panel of the ramp group leaves
# behind ITS last launch (BOF_PANEL_RAMP_FLUSH=1, the new default) against the whole group leaving behind the group's
# last launch (=0), plus the writer count and the ramp group's size.  Interleaved rounds of
# `bench.py --no-cpu --no-extras --no-probe --steps N --warmup 2` through tools/r6/ab_headline.sh's free-form variants.
# Usage: tools/r6/ab_ramp_flush.sh OUTDIR [ROUNDS] [STEPS]
out=${1:-gpurun_out/ab_ramp}; rounds=${2:-4}; steps=${3:-10}
exec bash tools/r6/ab_headline.sh "$out" "$rounds" "$steps" \
  "old:BOF_PANEL_RAMP_FLUSH=0:--no-probe" \
  "new:BOF_PANEL_RAMP_FLUSH=1:--no-probe" \
  "new_w8:BOF_PANEL_RAMP_FLUSH=1,BOF_PANEL_WRITERS=8:--no-probe" \
  "new_g3:BOF_PANEL_RAMP_FLUSH=1,BOF_PANEL_GROUP=3:--no-probe" \
  "new_g2w8:BOF_PANEL_RAMP_FLUSH=1,BOF_PANEL_GROUP=2,BOF_PANEL_WRITERS=8:--no-probe"
