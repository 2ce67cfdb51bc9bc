set -o pipefail
tools/gpu_check.sh || exit 1
tools/rehearse_n.sh 5 1000003 2>&1 | tail -2 || exit 1
# config 5 at its own 1e8 frames, one rank over RCCL (group of one), the line kept as a record
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$(python3 -c "import socket; s=socket.socket(); s.bind((\"127.0.0.1\", 0)); print(s.getsockname()[1])") BENCH_GROUP_OF_ONE=1 BENCH_TEST_HOOKS=1 timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 1 --frames 1000000 --placements 1 --sustain-s 0 --config5-frames 100000000 --leg-timeout 600 > gpurun_out/config5_1e8_one_rank.json 2> gpurun_out/config5_1e8_one_rank.err
echo "config5 1e8 rc $?"; cut -c1-600 gpurun_out/config5_1e8_one_rank.json | tail -1
