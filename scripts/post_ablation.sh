#!/bin/bash
# k_post's duration with pieces switched off (KLNMF_POST_ABL bits: 1 no normalisation pass, 2 no slab pass, 4 no last-block
# counter, 8 no loss / stop rule): where its time goes.  Usage (GPU box): bash scripts/post_ablation.sh "<bench args>"
R=${GRAFT_REPO_ROOT:-$(pwd)}
for abl in 0 1 2 4 8 15; do
  KLNMF_POST_ABL=$abl python3 $R/scripts/timeline.py $R/gpurun_out/post_abl_$abl -- $1 --steps 20 --warmup 5 --repeats 1 --data device --no-cpu-baseline --no-16bit-segment 2>&1 | grep -E "k_post" | sed "s/^/abl=$abl  /"
done
