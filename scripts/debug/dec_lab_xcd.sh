# ON THE GPU BOX: persistent decoder step, batch 512: ordinary stores inside a cluster that sits on one XCD (default) against
# write-through stores (TPSPP_HEAD_WRITE_THROUGH=1: the first version of the kernel), one launch per step / per decode
python scripts/debug/bench_decoder_modes.py 0 15 2>&1 | grep -v "^/opt" | tail -3
echo "write-through stores"; TPSPP_HEAD_WRITE_THROUGH=1 python scripts/debug/bench_decoder_modes.py 0 15 2>&1 | grep -v "^/opt" | tail -3
