#!/usr/bin/env python3
"""A/B of the LDS-staged message reads of the curve-hash accumulate kernel (round 4).  Run once per arm:
    AKP_TE_MSG_LDS=0 python tools/gpu_te_msg_lds.py     # per-lane 32-bit global loads at a 128-byte pitch (round 3)
    AKP_TE_MSG_LDS=1 python tools/gpu_te_msg_lds.py     # messages staged through LDS once per workgroup (default)
Legs: the 2 x 2 of VERDICT r03 weak #3 -- {random, one} table index x {global per-lane, LDS} message reads -- for Pedersen
4x256 / 128 B and Bowe-Hopwood 63x9 / 64 B at 2^20 messages ("one message" is 2^20 copies at distinct addresses: every
lane reads its own bytes but selects the same table entry); a Bowe-Hopwood 2^20-leaf tree; the host-pointer entry point
with pageable and pinned buffers (pinned input is zero copy when the LDS kernel runs)."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen as cped, bowe_hopwood as cbh  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
st = torch.cuda.current_stream().cuda_stream
n = 1 << 20
arm = os.environ.get("AKP_TE_MSG_LDS", "1")
print("# arm AKP_TE_MSG_LDS=%s" % arm)


def timed(fn, reps=15, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    return ms[len(ms) // 2]


rng = np.random.default_rng(1)
Pp = cped.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256))
Pb = cbh.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9))
for name, P, L, fe in (("pedersen 4x256, 128 B", Pp, 128, 8), ("bowe-hopwood 63x9, 64 B", Pb, 64, 4), ("bowe-hopwood 63x9, 32 B", Pb, 32, 4)):
    h = P.handle(ctx)
    out = torch.empty((n, fe), dtype=torch.int64, device=dev)
    rnd = torch.from_numpy(rng.integers(0, 256, size=(n, L), dtype=np.uint8)).to(dev)
    same = rnd[:1].repeat(n, 1).contiguous()
    few = rnd[:64].repeat(n // 64, 1).contiguous()
    for label, m in (("random messages", rnd), ("64 distinct messages", few), ("one message", same)):
        ms = timed(lambda: check(lib.akp_te_crh_batch_dev(h.h, m.data_ptr(), n, L, out.data_ptr(), st)))
        print("%-26s %-22s %.3f ms  %.4g hashes/s" % (name, label, ms, n / ms * 1e3))

# Bowe-Hopwood tree, 2^20 and 2^23 leaves of 32 bytes, resident
hb = Pb.handle(ctx)
for lg in (20, 23):
    nl = 1 << lg
    leaves = torch.from_numpy(rng.integers(0, 256, size=(nl, 32), dtype=np.uint8)).to(dev)
    ln = torch.empty((nl, 4), dtype=torch.int64, device=dev)
    nn = torch.empty((nl - 1, 4), dtype=torch.int64, device=dev)
    ms = timed(lambda: check(lib.akp_merkle_build_te_dev(hb.h, hb.h, leaves.data_ptr(), nl, 32, ln.data_ptr(), nn.data_ptr(), st)), reps=7, warm=2)
    print("bowe-hopwood tree 2^%d leaves %.3f ms  %.4g leaves/s" % (lg, ms, nl / ms * 1e3))
    del leaves, ln, nn

# host-pointer entry point: pageable vs pinned (input and output)
hp = Pp.handle(ctx)
hm = np.random.default_rng(0xA5A50014).integers(0, 256, size=(n, 128), dtype=np.uint8)
ho = np.empty((n, 8), dtype=np.uint64)


def wall(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


s = wall(lambda: check(lib.akp_te_crh_batch(hp.h, hm.ctypes.data, n, 128, ho.ctypes.data)))
print("host path pedersen pageable in/out  %.3f ms  %.4g hashes/s" % (s * 1e3, n / s))
pm, po = C.c_void_p(), C.c_void_p()
check(lib.akp_host_alloc(hm.nbytes, C.byref(pm)))
check(lib.akp_host_alloc(ho.nbytes, C.byref(po)))
np.ctypeslib.as_array((C.c_uint8 * hm.size).from_address(pm.value))[:] = hm.reshape(-1)
s = wall(lambda: check(lib.akp_te_crh_batch(hp.h, pm, n, 128, po)))
pinned_out = np.ctypeslib.as_array((C.c_uint64 * ho.size).from_address(po.value)).reshape(ho.shape)
print("host path pedersen pinned in/out    %.3f ms  %.4g hashes/s  equal=%s" % (s * 1e3, n / s, bool(np.array_equal(pinned_out, ho))))
ho2 = np.empty_like(ho)
s = wall(lambda: check(lib.akp_te_crh_batch(hp.h, pm, n, 128, ho2.ctypes.data)))
print("host path pedersen pinned in, pageable out %.3f ms  %.4g hashes/s  equal=%s" % (s * 1e3, n / s, bool(np.array_equal(ho2, ho))))
s = wall(lambda: check(lib.akp_te_crh_batch(hp.h, hm.ctypes.data, n, 128, po)))
print("host path pedersen pageable in, pinned out %.3f ms  %.4g hashes/s" % (s * 1e3, n / s))
# Bowe-Hopwood leaves 32 B, pinned
hb32 = np.random.default_rng(7).integers(0, 256, size=(n, 32), dtype=np.uint8)
hbo = np.empty((n, 4), dtype=np.uint64)
s = wall(lambda: check(lib.akp_te_crh_batch(hb.h, hb32.ctypes.data, n, 32, hbo.ctypes.data)))
print("host path bowe-hopwood 32 B pageable %.3f ms  %.4g hashes/s" % (s * 1e3, n / s))
pb = C.c_void_p(); pbo = C.c_void_p()
check(lib.akp_host_alloc(hb32.nbytes, C.byref(pb))); check(lib.akp_host_alloc(hbo.nbytes, C.byref(pbo)))
np.ctypeslib.as_array((C.c_uint8 * hb32.size).from_address(pb.value))[:] = hb32.reshape(-1)
s = wall(lambda: check(lib.akp_te_crh_batch(hb.h, pb, n, 32, pbo)))
pbout = np.ctypeslib.as_array((C.c_uint64 * hbo.size).from_address(pbo.value)).reshape(hbo.shape)
print("host path bowe-hopwood 32 B pinned   %.3f ms  %.4g hashes/s  equal=%s" % (s * 1e3, n / s, bool(np.array_equal(pbout, hbo))))
