# Round 4, VERDICT r03 item 2a: the decode attention pair (attn_decode + combine), two experiments with the kill criterion stated first:
#   keep a change only if the pair's kernel time drops to <= 9.5 us per sliding-window layer (12.65 today) AND bf16 decode tok/s rises on the same box.
#   A. XCD-local partials (attn.xcd_local=1): splits of a head group + its combine workgroups on one XCD
#   B. splits x combine at HS 512 / 2K: attn.positions_per_split, attn.max_workgroups, attn.heads_per_group_512
# usage (GPU box): bash tools/experiments/attn_pair.sh  -> gpurun_out/attn_pair.txt
set -e
cd $GRAFT_REPO_ROOT
out=gpurun_out/attn_pair.txt
: > $out
run() {
  tag=$1; shift
  line=$(python3 bench.py --policies bf16 --no-prefill --no-cpu --no-gpt2 --steps 96 --warmup 16 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "$tag $line" | tee -a $out
}
run baseline
run xcd_local --tune attn.xcd_local=1
run baseline_again
run split32 --tune attn.positions_per_split=32
run split32_wg512 --tune attn.positions_per_split=32 --tune attn.max_workgroups=512
run split64_wg512 --tune attn.max_workgroups=512
run gh512_4 --tune attn.heads_per_group_512=4
run gh512_4_wg512 --tune attn.heads_per_group_512=4 --tune attn.max_workgroups=512
run xcd_local_wg512 --tune attn.xcd_local=1 --tune attn.max_workgroups=512
run xcd_local_gh4 --tune attn.xcd_local=1 --tune attn.heads_per_group_512=4
run baseline_last
