#!/bin/bash
# round 5: the profile session behind profiles/r05_* (retune, forward bench + kernel stats + counter passes incl. 1024x320,
# training profiles + counters, sparse workloads + timelines, GPU suite), stamps, and the 2-rank rehearsals of bench.py
bash tools/profile_session.sh r05 retune fwd bwd sparse tests
OUT=$PWD/gpurun_out/r05
WMD_LIB_PATH=$PWD/tools/probes/_build/libwmd_stamps.so timeout 300 python tools/probes/stamps_probe.py > $OUT/stamps.txt 2>&1
# N = 2 launch contract on one GPU (ranks share the device, collectives through gloo): the headline + train extras
WMD_BENCH_BACKEND=gloo WMD_BENCH_SHARE_DEVICES=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --train-steps 3 --exchange-backend torch --no-train-nyu > $OUT/bench_2rank_rehearsal.json 2> $OUT/bench_2rank_rehearsal.err
tail -c 700 $OUT/bench_2rank_rehearsal.json; echo
# ... and with the RCCL exchange, which cannot come up between two ranks on ONE device: the line must carry every rank's error
WMD_BENCH_BACKEND=gloo WMD_BENCH_SHARE_DEVICES=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --train-steps 3 --exchange-backend rccl --no-train-nyu > $OUT/bench_2rank_rccl_failure.json 2> $OUT/bench_2rank_rccl_failure.err
python -c "import json; d=json.loads([l for l in open('$OUT/bench_2rank_rccl_failure.json') if l.startswith('{')][-1]); print('train:', json.dumps(d['train'])[:900])"   # (RCCL / gloo banners precede the line)
