#!/bin/bash
# usage (one-GPU box, repo root): tools/rehearse_n.sh [N=5] [config-5 frames=1000003]  -- bench.py's N > 1 control flow with N fresh rank processes on
# the ONE device (BENCH_BACKEND=gloo BENCH_SINGLE_DEVICE=1: pieces staged through the host; RCCL refuses two ranks on one device), uneven tails, >= 3
# gather rounds per rank; the root's checksum over all gathered columns against ONE process rendering the same stream (tools/config5_single.py).
# N is at most 5 here: a GPU box of this pool admits six processes on its card and torch.distributed.run's agent is one of them (six ranks: the process guard ended the run) (N = 8 runs under gloo with a stub engine in tests/test_bench_config5.py).
n=${1:-5}; total=${2:-1000003}
[ "$n" -le 5 ] || { echo "at most 5 ranks on one card (the launcher's agent process counts as the sixth)"; exit 2; }
mkdir -p gpurun_out
BENCH_BACKEND=gloo BENCH_SINGLE_DEVICE=1 timeout -k 10 600 python bench.py --gpus $n --steps 2 --warmup 1 --frames 65536 --placements 1 --sustain-s 0 \
    --config5-frames $total --config5-chunk 32768 --leg-timeout 300 > gpurun_out/rehearsal_n$n.json 2> gpurun_out/rehearsal_n$n.err || { tail -5 gpurun_out/rehearsal_n$n.err; exit 1; }
single=$(timeout -k 10 300 python tools/config5_single.py $total 50000 | tail -1) || exit 1
python3 - "$n" "$total" "$single" <<'PY'
import json, sys
n, total, single = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
line = json.loads(open(f"gpurun_out/rehearsal_n{n}.json").read().strip().splitlines()[-1])
c5 = line["config5"]
ok = (c5["ranks_seen"] == list(range(n)) and c5["frames_total"] == total and sum(c5["frames_per_gpu"]) == total and c5["checksum_all_columns"] == single
      and c5["sharded_equals_single_gpu_on_first_chunk_of_every_rank"] and "error" not in line and line["n_gpus"] == n)
line["rehearsal"] = {"what": f"{n} rank processes on ONE MI355X over gloo (BENCH_SINGLE_DEVICE=1), bench.py unchanged otherwise", "single_process_checksum": single,
                     "checksum_equal": c5["checksum_all_columns"] == single, "rounds_per_rank": c5["rounds"], "ok": bool(ok)}
json.dump(line, open(f"gpurun_out/rehearsal_n{n}.json", "w"))
print("rehearsal", "ok" if ok else "FAILED", {k: c5[k] for k in ("ranks_seen", "frames_per_gpu", "rounds", "gathered_bytes", "checksum_all_columns")}, "single", single)
sys.exit(0 if ok else 1)
PY
