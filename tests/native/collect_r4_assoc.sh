# round-4 association numbers: Sinkhorn kernel alone, per-phase wall clock of a frame, kernel trace of the loop, lane-exchange probe
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -I $R/odam_amd/csrc $R/tests/native/sk_wave_probe.hip -o /tmp/sk_wave_probe 2>/dev/null && /tmp/sk_wave_probe > $O/r4_sk_wave_probe.txt 2>&1
python3 $R/tests/native/perf_sinkhorn.py 2>/dev/null > $O/r4_sinkhorn_alone.txt
python3 $R/tests/native/perf_assoc_phases.py 2>/dev/null > $O/r4_assoc_phases.txt
python3 $R/tests/native/perf_assoc_loop.py 2>/dev/null | head -30 > $O/r4_perf_assoc_loop.txt
bash $R/tests/native/prof_assoc.sh > $O/r4_assoc_kernels.txt 2>&1
python3 $R/tests/native/cmp_gnn_variants.py 2>/dev/null > $O/r4_gnn_variants.txt
python3 $R/tests/native/soak_assoc.py 300 40 2>/dev/null | tail -1 > $O/r4_soak_assoc_line.txt
