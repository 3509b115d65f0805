#!/usr/bin/env python3
"""What limits the host-pointer entry points?  Times, on this box: pinned / pageable H2D and D2H alone, both directions at
once on two streams, a copy under a running permutation kernel, and akp_poseidon_permute_batch at several chunk sizes
(AKP_HOST_CHUNK_LOG2 is read once per process, so each setting runs in a child process)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def t_ms(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        import ctypes as C
        import crypto_primitives_amd as cpa
        from crypto_primitives_amd import field
        from crypto_primitives_amd._lib import lib, check
        n = 1 << int(sys.argv[2])
        cfg = cpa.get_default_poseidon_parameters(2, False)
        ph = cfg.handle()
        st = field.random_fr(n * 3, seed=1).reshape(n, 3, 4)
        pp = C.c_void_p()
        check(lib.akp_host_alloc(st.nbytes, C.byref(pp)))
        arr = np.ctypeslib.as_array((C.c_uint64 * st.size).from_address(pp.value))
        arr[:] = st.reshape(-1)
        for label, ptr in (("pageable", st.ctypes.data), ("pinned", pp)):
            check(lib.akp_poseidon_permute_batch(ph.h, ptr, n))
            t0 = time.perf_counter()
            for _ in range(5):
                check(lib.akp_poseidon_permute_batch(ph.h, ptr, n))
            ms = (time.perf_counter() - t0) / 5 * 1e3
            print("  chunk 2^%s  n 2^%s  %-8s %7.2f ms  %6.1f M perm/s  %5.1f GB/s each way" % (os.environ.get("AKP_HOST_CHUNK_LOG2", "18"), sys.argv[2], label, ms, n / ms / 1e3, 96.0 * n / ms / 1e6))
        return
    dev = torch.device("cuda", 0)
    nbytes = 100 << 20
    hp = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    hq = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    hpage = torch.empty(nbytes, dtype=torch.uint8)
    d1 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d2 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    print("100 MiB copies:")
    ms = t_ms(lambda: d1.copy_(hp, non_blocking=True)); print("  H2D pinned    %6.2f ms  %5.1f GB/s" % (ms, nbytes / ms / 1e6))
    ms = t_ms(lambda: hq.copy_(d2, non_blocking=True)); print("  D2H pinned    %6.2f ms  %5.1f GB/s" % (ms, nbytes / ms / 1e6))
    ms = t_ms(lambda: d1.copy_(hpage)); print("  H2D pageable  %6.2f ms  %5.1f GB/s" % (ms, nbytes / ms / 1e6))
    ms = t_ms(lambda: hpage.copy_(d2)); print("  D2H pageable  %6.2f ms  %5.1f GB/s" % (ms, nbytes / ms / 1e6))

    def both():
        with torch.cuda.stream(s1):
            d1.copy_(hp, non_blocking=True)
        with torch.cuda.stream(s2):
            hq.copy_(d2, non_blocking=True)
    ms = t_ms(both); print("  H2D + D2H on two streams  %6.2f ms  (%5.1f GB/s each way)" % (ms, nbytes / ms / 1e6))
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    from crypto_primitives_amd._lib import lib, check
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle()
    n = 1 << 20
    st = torch.from_numpy(field.random_fr(n * 3, seed=1).reshape(n, 3, 4).view(np.int64)).to(dev)

    def kern():
        check(lib.akp_poseidon_permute_batch_dev(ph.h, st.data_ptr(), n, s1.cuda_stream))
    ms = t_ms(kern); print("  permutation kernel 2^20 alone %6.2f ms" % ms)

    def kern_and_copy():
        kern()
        with torch.cuda.stream(s2):
            d1.copy_(hp, non_blocking=True)
    ms = t_ms(kern_and_copy); print("  kernel (s1) + H2D pinned (s2)  %6.2f ms" % ms)

    def kern_and_both():
        kern()
        with torch.cuda.stream(s2):
            d1.copy_(hp, non_blocking=True)
            hq.copy_(d2, non_blocking=True)
    ms = t_ms(kern_and_both); print("  kernel (s1) + H2D, D2H pinned (s2)  %6.2f ms" % ms)
    for lg in ("16", "17", "18", "19", "20"):
        subprocess.call([sys.executable, os.path.abspath(__file__), "child", "20"], env=dict(os.environ, AKP_HOST_CHUNK_LOG2=lg))
    subprocess.call([sys.executable, os.path.abspath(__file__), "child", "22"], env=dict(os.environ, AKP_HOST_CHUNK_LOG2="18"))
    for v in ("0", "1"):
        print("HSA_ENABLE_SDMA=%s:" % v)
        subprocess.call([sys.executable, os.path.abspath(__file__), "child", "20"], env=dict(os.environ, AKP_HOST_CHUNK_LOG2="18", HSA_ENABLE_SDMA=v))


if __name__ == "__main__":
    main()
