"""The pinned host path of the curve hashes as ONE gated launch (round 5, capi_te.hip te_crh_gated): the accumulate kernel covers the
whole batch while its messages arrive by DMA chunk after chunk; workgroups wait on arrival flags, the host thread releases each chunk's
finalize pass and copy-out.  Digests against the oracle and against the pageable call, over batch shapes that exercise partial
chunks / partial workgroups, repeated calls (the epoch that tags one call's flags) and both hash kinds."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, cref  # noqa: E402
from helpers import gens_array  # noqa: E402


@pytest.fixture(autouse=True)
def _pinned_calls_take_the_gated_launch(monkeypatch):
    """round 6: the library measures which form serves a context's pinned batches faster (last test of this file); the tests of the gated
    launch itself pin the form"""
    monkeypatch.setenv("AKP_TE_PINNED_FORM", "gated")


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


class _Pinned:
    def __init__(self, cpa, nbytes):
        self.cpa, self.p = cpa, C.c_void_p()
        cpa._lib.check(cpa.lib.akp_host_alloc(nbytes, C.byref(self.p)))
        self.nbytes = nbytes

    def array(self, dtype, shape):
        n = int(np.prod(shape))
        ct = {np.uint8: C.c_uint8, np.uint64: C.c_uint64}[dtype]
        return np.ctypeslib.as_array((ct * n).from_address(self.p.value)).reshape(shape)

    def free(self):
        self.cpa._lib.check(self.cpa.lib.akp_host_free(self.p))


@pytest.mark.parametrize("kind,W,N,L", [("pedersen", 4, 256, 128), ("bh", 63, 9, 64), ("bh", 63, 9, 32), ("pedersen", 4, 256, 40)])
def test_pinned_batches_through_the_gated_launch(cpa, kind, W, N, L):
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    if kind == "pedersen":
        g = gens_array(jj.pedersen_generators(0xE5E50001, W, N))
        prm, fe = pedersen.Parameters(g), 2
        ora = lambda m, k: cref.CurveParams(W, N, g).pedersen_crh_batch(m, k, L, threads=8)  # noqa: E731
    else:
        g = gens_array(jj.bowe_hopwood_generators(0xE5E50002, W, N))
        prm, fe = bowe_hopwood.Parameters(g), 1
        ora = lambda m, k: cref.CurveParams(W, N, g).bh_crh_batch(m, k, L, threads=8)  # noqa: E731
    h = prm.handle()
    chunk = 1 << 17
    # 8, 13, 9 and 20 granules of 2^15 messages: uniform chunks below 12 granules, the ramped schedule (1 1 2 4 .. 2 1 1) from there;
    # for the 32-byte Bowe-Hopwood case also 2^22 + 5 messages (granules of 2^17: 33 chunks) and 2^23 + 2^16 (granules of 2^18, 33 chunks)
    sizes = (2 * chunk, 3 * chunk + 1000, 2 * chunk + 1, 5 * chunk - 255) + (((1 << 22) + 5, (1 << 23) + (1 << 16)) if L == 32 else ())
    for rep, n in enumerate(sizes):
        msgs = np.random.default_rng(100 * rep + L).integers(0, 256, size=(n, L), dtype=np.uint8)
        want = np.empty((n, 4 * fe), np.uint64)
        cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, msgs.ctypes.data, n, L, want.ctypes.data))  # pageable: the chunked launches
        pm, po = _Pinned(cpa, msgs.nbytes), _Pinned(cpa, want.nbytes)
        try:
            pm.array(np.uint8, msgs.shape)[:] = msgs
            out = po.array(np.uint64, want.shape)
            for again in range(2):  # twice: the second call's flags carry the next epoch
                out[:] = 0
                cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, pm.p, n, L, po.p))
                assert np.array_equal(out, want), (kind, L, n, again, np.nonzero((out != want).any(axis=1))[0][:8])
        finally:
            pm.free()
            po.free()
        si = np.unique(np.concatenate([np.arange(40), np.arange(chunk - 20, chunk + 20), np.linspace(0, n - 1, 300).astype(np.int64), np.arange(n - 40, n)]))
        assert np.array_equal(want[si].reshape(len(si), fe, 4), ora(np.ascontiguousarray(msgs[si]), len(si)).reshape(len(si), fe, 4)), (kind, L, n)


@pytest.mark.parametrize("kind,W,N,L", [("pedersen", 4, 256, 128), ("bh", 63, 9, 64)])
def test_consecutive_gated_calls_read_the_bytes_of_their_own_batch(cpa, kind, W, N, L):
    """The workgroups of a gated launch read bytes that a copy engine wrote WHILE the kernel was running, into a device buffer that the
    previous call's kernel read at the same addresses: a cache line of the previous batch that survived in an L2 would go unnoticed if
    both calls hashed the same messages.  Batches A and B (same shape, different bytes) alternate through the same pinned buffer with no
    other call between them; expected digests from the pageable path, computed beforehand."""
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    if kind == "pedersen":
        prm, fe = pedersen.Parameters(gens_array(jj.pedersen_generators(0xE5E50001, W, N))), 2
    else:
        prm, fe = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0xE5E50002, W, N))), 1
    h = prm.handle()
    n = (1 << 18) + 4096  # 32 MB of 128-byte messages: what the eight L2s and the memory-side cache can hold
    batches, wants = [], []
    for seed in (1, 2):
        m = np.random.default_rng(seed).integers(0, 256, size=(n, L), dtype=np.uint8)
        w = np.empty((n, 4 * fe), np.uint64)
        cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, n, L, w.ctypes.data))
        batches.append(m)
        wants.append(w)
    pm, po = _Pinned(cpa, batches[0].nbytes), _Pinned(cpa, wants[0].nbytes)
    try:
        out = po.array(np.uint64, wants[0].shape)
        for turn in (0, 1, 0, 1, 1, 0):
            pm.array(np.uint8, batches[turn].shape)[:] = batches[turn]
            out[:] = 0
            cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, pm.p, n, L, po.p))
            assert np.array_equal(out, wants[turn]), (kind, turn, np.nonzero((out != wants[turn]).any(axis=1))[0][:8])
    finally:
        pm.free()
        po.free()


def test_two_threads_with_contexts_of_their_own_hash_pinned_batches_at_once(cpa):
    """the shim's shape: every OS thread has a context; two threads push pinned batches through the same parameters at the same time.
    Only ONE gated launch runs per device (the other caller takes the chunked launches: two gated kernels would hold every wave slot
    with waiting workgroups); both get the oracle's digests, call after call, and the table is shared"""
    import threading
    from crypto_primitives_amd._lib import Context
    from crypto_primitives_amd.crh import bowe_hopwood
    g = gens_array(jj.bowe_hopwood_generators(0xE5E50003, 63, 9))
    ora = cref.CurveParams(63, 9, g)
    B = bowe_hopwood.Parameters(g)
    ctxs = [Context(0), Context(0)]
    n, L = (1 << 18) + 77, 64
    errs = []

    def work(i):
        try:
            h = B.handle(ctxs[i])
            msgs = np.random.default_rng(50 + i).integers(0, 256, size=(n, L), dtype=np.uint8)
            pm, po = _Pinned(cpa, msgs.nbytes), _Pinned(cpa, n * 32)
            pm.array(np.uint8, msgs.shape)[:] = msgs
            out = po.array(np.uint64, (n, 4))
            si = np.unique(np.concatenate([np.arange(64), np.linspace(0, n - 1, 400).astype(np.int64), np.arange(n - 64, n)]))
            want = ora.bh_crh_batch(np.ascontiguousarray(msgs[si]), len(si), L, threads=2)
            first = None
            for rep in range(6):
                out[:] = 0
                cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, pm.p, n, L, po.p))
                assert np.array_equal(out[si], want), (i, rep)
                if first is None:
                    first = out.copy()
                assert np.array_equal(out, first), (i, rep)
            pm.free()
            po.free()
        except BaseException as e:  # noqa: BLE001
            errs.append((i, repr(e)))
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert B.handle(ctxs[0]).table_info()["table_id"] == B.handle(ctxs[1]).table_info()["table_id"]


def test_a_gate_that_fails_falls_back_to_the_chunked_launches():
    """the safety net of the gated launch: with a spin limit of ONE poll (test build, AKP_TE_GATE_SPIN_LIMIT=1) every workgroup whose chunk
    has not arrived yet gives up at once; the call must notice (error word), repeat the batch with round 4's chunked launches, return the
    right digests, and the context must back off (the next 8 pinned calls go straight to the chunked launches, the 9th tries the gate again)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = os.path.join(root, "crypto_primitives_amd", "lib", "libakp_testhooks.so")
    assert os.path.exists(hooks)
    code = r"""
import ctypes as C, sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import crypto_primitives_amd as cpa
from crypto_primitives_amd.crh import bowe_hopwood
from oracle import jubjub as jj, cref
from helpers import gens_array
g = gens_array(jj.bowe_hopwood_generators(0xE5E50009, 63, 9))
B = bowe_hopwood.Parameters(g)
h = B.handle()
n, L = (1 << 18) + 5, 64
msgs = np.random.default_rng(1).integers(0, 256, size=(n, L), dtype=np.uint8)
want = np.empty((n, 4), np.uint64)
cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, msgs.ctypes.data, n, L, want.ctypes.data))
pm, po = C.c_void_p(), C.c_void_p()
cpa._lib.check(cpa.lib.akp_host_alloc(msgs.nbytes, C.byref(pm))); cpa._lib.check(cpa.lib.akp_host_alloc(want.nbytes, C.byref(po)))
np.ctypeslib.as_array((C.c_uint8 * msgs.size).from_address(pm.value))[:] = msgs.reshape(-1)
out = np.ctypeslib.as_array((C.c_uint64 * want.size).from_address(po.value)).reshape(want.shape)
times, notes = [], []
for rep in range(10):
    out[:] = 0
    cpa.lib.akp_te_params_prepare(None, 0)  # (sets the thread's last error to something else)
    t0 = time.perf_counter()
    cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, pm, n, L, po))
    times.append(time.perf_counter() - t0)
    notes.append(cpa.lib.akp_last_error().decode())
    assert np.array_equal(out, want), rep
# a timeout is a moment, not a property of the stack (ADVICE r05): call 0 tried the gate and says so, calls 1..8 back off (chunked launches
# straight away), call 9 tries again
assert "gated launch" in notes[0] and "1 time(s)" in notes[0] and "next 8 pinned calls" in notes[0], notes[0]
assert all("gated launch" not in x for x in notes[1:9]), notes
assert "2 time(s)" in notes[9] and "next 16 pinned calls" in notes[9], notes[9]
si = np.linspace(0, n - 1, 200).astype(np.int64)
assert np.array_equal(want[si], cref.CurveParams(63, 9, g).bh_crh_batch(np.ascontiguousarray(msgs[si]), len(si), L, threads=4))
print("FALLBACK OK", ["%%.1f ms" %% (t * 1e3) for t in times])
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, AKP_LIB=hooks, AKP_TE_GATE_SPIN_LIMIT="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "FALLBACK OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])


def test_the_form_of_a_pinned_call_is_measured_not_assumed():
    """round 6 (profiles/r06_s41 ... s46): where the runtime puts the copy streams decides whether the gated launch or the chunked launches
    serve a pinned batch faster, so the context measures: four calls of either form in turns, then the faster form with every 32nd call
    given to the other; a new message length starts over; with the gate switched off (test build) every call takes the chunked launches.
    The test build reports which form ran each call (AKP_TE_GATE_REPORT); digests are checked on every call."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hooks = os.path.join(root, "crypto_primitives_amd", "lib", "libakp_testhooks.so")
    assert os.path.exists(hooks)
    code = r"""
import ctypes as C, os, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import crypto_primitives_amd as cpa
from crypto_primitives_amd.crh import bowe_hopwood
from oracle import jubjub as jj
from helpers import gens_array
B = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0xE5E5000A, 63, 9)))
h = B.handle()
n = (1 << 18) + 5
report = os.environ["AKP_TE_GATE_REPORT"]
def forms():
    return [int(l.split()[2]) for l in open(report)] if os.path.exists(report) else []
major_of = {}
for L, calls in ((64, 70), (32, 6), (64, 1)):  # (the third: back to the first shape -- its figures are still there, no new round of turns)
    msgs = np.random.default_rng(L).integers(0, 256, size=(n, L), dtype=np.uint8)
    want = np.empty((n, 4), np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, msgs.ctypes.data, n, L, want.ctypes.data))  # pageable: not a pinned call, not reported
    pm, po = C.c_void_p(), C.c_void_p()
    cpa._lib.check(cpa.lib.akp_host_alloc(msgs.nbytes, C.byref(pm))); cpa._lib.check(cpa.lib.akp_host_alloc(want.nbytes, C.byref(po)))
    np.ctypeslib.as_array((C.c_uint8 * msgs.size).from_address(pm.value))[:] = msgs.reshape(-1)
    out = np.ctypeslib.as_array((C.c_uint64 * want.size).from_address(po.value)).reshape(want.shape)
    before = len(forms())
    notes = []
    for rep in range(calls):
        out[:] = 0
        cpa.lib.akp_te_params_prepare(None, 0)  # (sets the thread's last error to something else)
        cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, pm, n, L, po))
        notes.append(cpa.lib.akp_last_error().decode())
        assert np.array_equal(out, want), (L, rep)
    if calls > 8 and os.environ.get("AKP_TE_GATED") != "0":  # the choice is announced behind the call that made it (the ninth), and only changes of it later
        assert notes[8].startswith("note: the pinned curve-hash batches of this context take the "), notes[8]
        assert not any(x.startswith("note:") for x in notes[:8]), notes[:8]
    f = forms()[before:]
    assert len(f) == calls, f
    if os.environ.get("AKP_TE_GATED") == "0":
        assert not any(f), f
    else:
        if calls == 1:
            assert f == [major_of[L]], (f, major_of)
        else:
            assert f[:8] == [1, 0, 1, 0, 1, 0, 1, 0][:calls], f   # four of either form first, in turns -- also for the second message length
        if calls > 40:
            steady = f[8:]
            major = 1 if sum(steady) * 2 > len(steady) else 0
            major_of[L] = major
            other = [i for i, x in enumerate(steady) if x != major]
            assert 1 <= len(other) <= 3, f          # 62 calls: the other form is looked at again on every 32nd
    cpa._lib.check(cpa.lib.akp_host_free(pm)); cpa._lib.check(cpa.lib.akp_host_free(po))
print("TUNE OK", forms())
""" % (root, os.path.join(root, "tests"))
    for gate in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            env = dict(os.environ, AKP_LIB=hooks, AKP_TE_GATE_REPORT=os.path.join(d, "forms.txt"), AKP_TE_GATED=gate)
            env.pop("AKP_TE_PINNED_FORM", None)
            p = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
            assert p.returncode == 0 and "TUNE OK" in p.stdout, (gate, p.stdout[-1500:], p.stderr[-3000:])
