#!/bin/bash
# round 5, experiment 4: what would a tile cost if the previous tile's epilogue traffic rode under its products?  (timing-only builds, wrong results)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
(echo "== this build"; ./tests/native/ring_stamps0 x16; echo "== no epilogue at all (CG_HACK=1)"; ./tests/native/ring_stamps0_hack1 x16
 echo "== no epilogue, its loads and stores inside the main loop (CG_HACK=3)"; ./tests/native/ring_stamps0_hack3 x16
 echo "== this build again"; ./tests/native/ring_stamps0 x16; echo "== light stamps, one workgroup in 64 reporting"; ./tests/native/ring_stamps2 x16) > $O/r5_ring_hacks.txt 2>&1
cat $O/r5_ring_hacks.txt
