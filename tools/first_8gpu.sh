#!/bin/bash
# The A/B list DESIGN 6 wants from the FIRST run on an 8 x MI355X node (BASELINE configs[2]: VQ-VAE-EMA, data parallel,
# batch 64 = 8 windows per GPU; train.py:58-60, chassis.py:168-169,188-190).  No such node has been available to the
# build, so every choice below is priced, not measured: this script measures them in one sitting.  Each line
# is the bench line of one configuration - whole-job samples/s, ms per step and, per wait point, how long the compute
# stream actually stalls on a collective (data_parallel.exposed_collective_ms_per_step / by_wait_ms_per_step).
#
#   usage (on the node):  tools/first_8gpu.sh [outdir] [gpus]
#
#   1 sharded      reduce-scatter + sharded Adam + all-gather, decoder all-gather under the next encoder forward (default)
#   2 all-reduce   round 1's schedule: decoder gradients all-reduced under the encoder backward, head under the decoder's Adam
#   3 bf16 grads   the sharded schedule with the gradients reduce-scattered through a bf16 copy (half the xGMI bytes)
#   4 two groups   AEW_WGRAD_GROUP=10: the stack's weight gradients as two grouped launches, the upper layers' region
#                  exchanged ~2 ms before the decoder's backward ends (costs 0.44 ms on one GPU: pays only below ~50 GB/s)
#   5 merged pack  all weight layouts packed in one launch at the head of the forward (what a lone process does): the
#                  decoder's all-gather can then not run under the encoder forward - the price of giving that overlap up
#   6 scaling      the default at 1, 2, 4 GPUs (the driver's SCALE run does this as well)
O=${1:-gpurun_out/first_8gpu}; N=${2:-8}
mkdir -p $O
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # name, env..., -- bench args
  name=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python bench.py --gpus $N --steps 30 --warmup 5 --no-cpu-baseline --check-replicas "$@" > $O/$name.json 2> $O/$name.err
  python - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    o = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    dp = o.get("data_parallel") or {}
    print(f"{sys.argv[2]:14s} n_gpus {o['n_gpus']}  {o['ms_per_step']:.3f} ms/step  {o['value'] / 1e6:.2f} M samples/s  "
          f"exposed {dp.get('exposed_collective_ms_per_step')} ms/step  by wait {dp.get('by_wait_ms_per_step')}  "
          f"replica diff {dp.get('replica_param_max_diff')}")
except Exception as e:
    print(f"{sys.argv[2]:14s} FAILED ({type(e).__name__}: {e}); see the .err file")
PY
}
run sharded      AEW_DP_SHARDED=1 --
run all_reduce   AEW_DP_SHARDED=0 --
run bf16_grads   AEW_DP_SHARDED=1 AEW_DP_BF16_GRADS=1 --
run two_groups   AEW_DP_SHARDED=1 AEW_WGRAD_GROUP=10 --
run merged_pack  AEW_DP_SHARDED=1 -- --merge-packs 1
for n in 1 2 4; do
  [ $n -lt $N ] || continue
  python bench.py --gpus $n --steps 30 --warmup 5 --no-cpu-baseline > $O/scale_$n.json 2> $O/scale_$n.err
  python -c "import json,sys; o=json.loads([l for l in open('$O/scale_$n.json') if l.startswith('{')][-1]); print(f\"scale n_gpus {o['n_gpus']}  {o['ms_per_step']:.3f} ms/step  {o['value']/1e6:.2f} M samples/s\")"
done
