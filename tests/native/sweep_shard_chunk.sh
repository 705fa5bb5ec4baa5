# a 125-frame shard (one rank of eight): frames per forward x frames per detections download (the host's threshold / NMS / row packing of a chunk
# runs under the GPU's work on the next): bench.py step, two rounds
mkdir -p gpurun_out/r6_aa
for rep in 1 2; do for bc in "42 42" "42 126" "21 63" "21 42" "14 42" "32 64"; do set -- $bc; python bench.py --frames 125 --objects 4 --batch $1 --chunk $2 --steps 4 --warmup 2 --no-secondary --no-config4 --no-cpu-baseline --no-full-chain 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $1 chunk $2: %.1f ms per step, detection phase %.1f ms' % (d['ms_per_step'], d['roofline']['detection_phase_ms_per_step']))"; done; done | tee gpurun_out/r6_aa/chunk_sweep.txt
