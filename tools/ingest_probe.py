"""Where the ~2.6 ms go that a step costs when the next batch's ingest runs beside it (bench.py --extras: 22.6 ms against 20.0): the
default workload timed (A) on one resident batch, (B) on a ring of three resident batches (what a fresh batch costs on the GPU side:
chunk metadata, output buffers), (C) A with the host packing into the staging set beside it but no copy, (D) A with the H2D copies of
an already packed staging set but no host packing, (E) the consumer loop's form (pack + copy, fresh batch every step) - and the host
time one step's launches take to enqueue.  Scratch tool.  Usage: python tools/ingest_probe.py [out.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dotaclient_amd.engine import Engine, IncrementalPacker, pack_rollouts, device_empty  # noqa: E402
from dotaclient_amd import synth  # noqa: E402

dev = torch.device('cuda:0')
B, S, E, STEPS = 256, 256, 4, 15
eng = Engine('lstm', 256, 1, dev)
eng.load_state_dict(synth.init_state_dict(7, 'lstm', 256, 1))
rollouts = synth.make_rollouts(1000, [S] * B)
ring = [pack_rollouts(rollouts, S, dev) for _ in range(3)]
packer = IncrementalPacker(S, dev, expected_rows=B * S)
side = torch.cuda.Stream(device=dev)


def one_step(batch):
    chunks = eng.rollout_pass(batch, S)
    for _ in range(E):
        eng.train_epoch(chunks, 5e-5, 5e-4, 0.5)


def timed(body):
    for i in range(3):
        body(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        body(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


out = {}
out['A_resident_batch_ms'] = timed(lambda i: one_step(ring[0]))
out['B_ring_of_three_resident_batches_ms'] = timed(lambda i: one_step(ring[i % 3]))


def fresh_buffers(i):
    b = ring[i % 3]
    b._bufs.clear(); b._chunk_meta.clear()          # what a batch that was never seen has to build
    one_step(b)


out['B2_ring_with_per_batch_buffers_rebuilt_ms'] = timed(fresh_buffers)


def pack_only(i):
    one_step(ring[0])
    for d in rollouts:
        packer.add(d)
    packer._pack_pending()
    packer._begin()                                  # drop it: host packing without the copies


out['C_host_packing_beside_ms'] = timed(pack_only)
st = packer.pair.take(B * S)
dst = [device_empty(x[:B * S].shape, x.dtype, dev) for x in (st.obs, st.act, st.msk, st.rew)]


def copy_only(i):
    one_step(ring[0])
    with torch.cuda.stream(side):
        for t, x in zip(dst, (st.obs, st.act, st.msk, st.rew)):
            t.copy_(x[:B * S], non_blocking=True)


out['D_h2d_copies_beside_ms'] = timed(copy_only)
cur = [ring[0]]


def full(i):
    one_step(cur[0])
    for d in rollouts:
        packer.add(d)
    cur[0] = packer.finish()


out['E_consumer_loop_form_ms'] = timed(full)
# host time of the enqueue alone (device idle at the start, nothing waits)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    one_step(ring[0])
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
out['host_enqueue_ms_of_one_step'] = ts
t0 = time.perf_counter()
for d in rollouts:
    packer.add(d)
packer._pack_pending()
out['host_pack_ms_of_one_batch'] = (time.perf_counter() - t0) * 1e3
packer._begin()
assert int(eng.status.item()) == 0
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
