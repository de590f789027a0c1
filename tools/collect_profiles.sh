#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root through gpurun):
#     gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r02'
# Writes into gpurun_out/<tag>/ ; copy what should be judged into profiles/ afterwards.
set -u
TAG=${1:-rXX}
REPO=$PWD
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python -c "import bench; print(bench.csrc_fingerprint())" > "$OUT/csrc_fingerprint.txt"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-workloads > "$OUT/bench_driver_style_20steps.json" 2>/dev/null
for wl in bibtex delicious; do
  python bench.py --workload $wl --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/bench_$wl.json" 2>/dev/null
done
python bench.py --workload synthetic4096 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_synthetic4096.json" 2>/dev/null
python bench.py --ragged --no-cpu-baseline > "$OUT/bench_reuters_ragged.json" 2>/dev/null
python bench.py --workload synthetic4096 --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined > "$OUT/bench_synthetic4096_b1024.json" 2>/dev/null
python bench.py --workload synthetic4096 --mask none --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined > "$OUT/bench_synthetic4096_none.json" 2>/dev/null
LAMP_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 5 --no-pipelined > "$OUT/bench_two_ranks_one_gpu_gloo.json" 2>/dev/null
LAMP_BENCH_BACKEND=gloo python bench.py --gpus 2 --ragged --steps 50 --warmup 5 --no-pipelined > "$OUT/bench_two_ranks_one_gpu_gloo_ragged.json" 2>/dev/null
python tools/check_rccl_control_plane.py > "$OUT/rccl_control_plane_one_rank.txt" 2>&1
python tools/bench_eval_epoch.py > "$OUT/eval_epoch_end_to_end.json" 2>/dev/null
python tools/bench_kernels.py gemm_ab 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_tiles.txt"
python tools/bench_kernels.py gemm 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_tiles_sweep.txt"
python tools/bench_kernels.py attn 2>&1 | grep -v amdgpu.ids > "$OUT/attn_variants.txt"
python tools/bench_kernels.py sparse_rows 2>&1 | grep -v amdgpu.ids > "$OUT/sparse_label_attention.txt"
python tools/bench_kernels.py sparse 2>&1 | grep -v amdgpu.ids >> "$OUT/sparse_label_attention.txt"
python bench.py --workload synthetic4096 --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-sparse-label-attention > "$OUT/bench_synthetic4096_dense_label_attention.json" 2>/dev/null
# same-box A/B of the weights-only embedding fold (round 6), three round-robin rounds per workload
AB_STEPS=200 bash tools/ab_flags.sh "$TAG" "" "" "--no-embed-fold" > /dev/null
AB_STEPS=200 bash tools/ab_flags.sh "$TAG" "--ragged" "" "--no-embed-fold" > /dev/null
AB_STEPS=100 bash tools/ab_flags.sh "$TAG" "--workload bibtex" "" "--no-embed-fold" > /dev/null
python tools/bench_kernels.py gemm_trace 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_trace.txt"
python tools/bench_kernels.py gemm_clock 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_clock.txt"
python tools/bench_kernels.py chain 2880 17 0,12,17 2>&1 | grep -v amdgpu.ids > "$OUT/chain.txt"
for m in 720 1440 2048 2400 2880 3072 3600 4096 4320 5120 5760 6144; do python tools/bench_kernels.py chain $m 0 0,12,15,16,17,19,20 2>&1 | grep "five\|chain launch" | tr "\n" ";" >> "$OUT/chain_rows.txt"; echo " M=$m" >> "$OUT/chain_rows.txt"; done
python tools/bench_kernels.py slab 2>&1 | grep -v amdgpu.ids > "$OUT/slab.txt"
tools/probes/mfma_korder > "$OUT/mfma_korder.txt" 2>&1
tools/probes/mfma_chain > "$OUT/mfma_chain_raw.txt" 2>&1
tools/probes/lds_dma_oob > "$OUT/lds_dma_oob.txt" 2>&1
python tools/bench_kernels.py attn_tile 2>&1 | grep -v amdgpu.ids > "$OUT/attn_tile.txt"
python tools/batch_sweep.py > "$OUT/batch_sweep.txt" 2>/dev/null
tools/probes/mfma_issue 20 > "$OUT/mfma_issue.txt" 2>&1
LAMP_BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 20 --warmup 5 --no-pipelined > "$OUT/bench_eight_ranks_one_gpu_gloo.json" 2>/dev/null
LAMP_BENCH_BACKEND=gloo python bench.py --gpus 8 --ragged --steps 20 --warmup 5 --no-pipelined > "$OUT/bench_eight_ranks_one_gpu_gloo_ragged.json" 2>/dev/null
BENCH="python $PWD/bench.py --no-cpu-baseline --no-pipelined --no-extra-workloads"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o p -f csv -- $BENCH --steps 100 --warmup 10 > "$OUT/bench_under_rocprof.json" 2>/dev/null )
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/stats_ragged" -o p -f csv -- $BENCH --ragged --steps 100 --warmup 10 > "$OUT/bench_ragged_under_rocprof.json" 2>/dev/null )
# PMC passes are separate runs, each with --kernel-trace only (never combined with other trace domains)
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
      -d "$OUT/pmc_sq" -o p -f csv -- $BENCH --steps 5 --warmup 3 > /dev/null 2>&1 )
for wl in reuters bibtex delicious; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && rocprofv3 --kernel-trace --pmc $c -d "$OUT/pmc_${c}_$wl" -o p -f csv -- $BENCH --workload $wl --steps 5 --warmup 3 > /dev/null 2>&1 )
  done
done
python tools/bench_train.py > "$OUT/train_reuters.json" 2>/dev/null
python tools/bench_train.py --per-launch --no-defer > "$OUT/train_reuters_per_launch_autograd_route.json" 2>/dev/null
python tools/bench_train.py --per-launch > "$OUT/train_reuters_per_launch_deferred.json" 2>/dev/null
python tools/bench_train.py --no-defer > "$OUT/train_reuters_composite_not_deferred.json" 2>/dev/null
python tools/bench_train.py --fused-adam > "$OUT/train_reuters_fused_adam.json" 2>/dev/null
python tools/bench_train.py --host-profile "$OUT/train_host_profile.txt" > /dev/null 2>&1
python tests/time_oracle_train_step.py > "$OUT/train_reuters_cpu_oracle.json" 2>/dev/null
python tools/bench_train.py --workload delicious --steps 5 --warmup 2 > "$OUT/train_delicious.json" 2>/dev/null
python tools/bench_kernels.py gemm_gen 2>&1 | grep -v amdgpu.ids > "$OUT/gemm_gen.txt"
( cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/train_stats" -o p -f csv -- python $REPO/tools/bench_train.py --steps 20 > /dev/null 2>&1 )
ls -R "$OUT" | head -40
