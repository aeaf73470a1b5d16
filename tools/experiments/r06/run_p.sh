#!/bin/bash
# round 6, GPU call p: every entry point alone at 13B widths, r = 64, 2 x 4096 tokens -- with the bench's token layout (3 of 32 token blocks hold two modalities)
# and with text only: do the two-walk workgroups of the token-owning kernels (one walk per modality of the run) set the launch time?
mkdir -p gpurun_out/r6p
export WIDTHS=5120x5120,5120x13824,13824x5120 R=${R:-64} B=2 S=4096 DROP=0.05 SWEEP=0
( echo "== bench layout"; timeout 600 python tools/tune_sweep.py; echo "== text only"; ALLTEXT=1 timeout 600 python tools/tune_sweep.py ) 2>&1 | grep -v "^$" | tee gpurun_out/r6p/entry_points_r$R.txt
