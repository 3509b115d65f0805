//! Per-thread device contexts, the wire-format check and error mapping.
use crate::ffi;
use crate::{Error, Fr};
use ark_ff::{BigInt, PrimeField};
use ark_std::{cell::RefCell, string::String, vec::Vec};
use core::ffi::CStr;

/// `Fr` must be 4 x u64 Montgomery limbs (`Fp(BigInt([u64; 4]), PhantomData)`), byte-identical to the ABI's wire
/// format.  ark-ff does not promise this layout, so it is asserted once per thread before any pointer cast.
pub fn layout_check() {
    assert_eq!(core::mem::size_of::<Fr>(), 32, "ark-ff changed the in-memory layout of Fp256");
    assert_eq!(core::mem::align_of::<Fr>(), 8);
    let one = Fr::from(1u64);
    // R mod p for BLS12-381 Fr
    assert_eq!((one.0).0, [0x0000_0001_ffff_fffe, 0x5884_b7fa_0003_4802, 0x998c_4fef_ecbc_4ff5, 0x1824_b159_acc5_056f]);
    // and the library agrees: canonical 1 -> Montgomery through the ABI
    let canon = [1u64, 0, 0, 0];
    let mut mont = [0u64; 4];
    let rc = unsafe { ffi::akp_fr_to_mont(canon.as_ptr(), mont.as_mut_ptr(), 1) };
    assert!(rc == ffi::AKP_OK && mont == (one.0).0, "libakp.so wire format differs from ark-ff's Fp256");
    assert_eq!(unsafe { ffi::akp_abi_version() }, ffi::AKP_ABI_VERSION, "libakp.so ABI version mismatch");
}

#[inline]
pub fn words(v: &[Fr]) -> *const u64 {
    v.as_ptr() as *const u64
}
#[inline]
pub fn words_mut(v: &mut [Fr]) -> *mut u64 {
    v.as_mut_ptr() as *mut u64
}
#[inline]
pub fn fr_from_limbs(l: [u64; 4]) -> Fr {
    Fr::new_unchecked(BigInt::new(l)) // the limbs are already the Montgomery representation
}

#[derive(Debug)]
pub struct AkpError(pub i32, pub String);
impl core::fmt::Display for AkpError {
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        write!(f, "libakp status {}: {}", self.0, self.1)
    }
}
impl ark_std::error::Error for AkpError {}

/// status code -> the reference's `Error` (`lib.rs:47-52`).  `len` is what `IncorrectInputLength` reports.
pub fn check(rc: i32, len: usize) -> Result<(), Error> {
    if rc == ffi::AKP_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(ffi::akp_last_error()) }.to_string_lossy().into_owned();
    match rc {
        // the reference panics on these lengths (crh/pedersen/mod.rs:82-89, crh/bowe_hopwood/mod.rs:121-129);
        // Poseidon has no length limit
        ffi::AKP_ERR_BAD_LENGTH => Err(Error::IncorrectInputLength(len)),
        // merkle_tree/mod.rs:430-433 asserts: keep the panic so callers see the reference's behaviour
        ffi::AKP_ERR_NOT_POW2 => panic!("`leaves.len() should be power of two and greater than one"),
        _ => Err(Error::GenericError(Box::new(AkpError(rc, msg)))),
    }
}

/// Handle cache of one thread: the traits hand us `&Parameters` on every call and a handle owns device tables worth up to
/// hundreds of MB, so handles are kept per distinct parameter set.  An entry stores the FULL key material next to the
/// handle: the 64-bit fingerprint only narrows the search, a hit needs `tag` and every field element to be equal -- a
/// fingerprint collision (trivial to construct for attacker-supplied or deserialised parameters) can therefore never hand out
/// another set's tables.  Bounded: beyond `cap` entries the least recently used handle is handed to `akp_*_params_destroy`,
/// so a thread that walks through many parameter sets does not keep their tables until it exits.
///
/// Eviction and objects that outlive the call: `GpuMerkleTree` and `GpuPoseidonSponge` keep the raw handle inside libakp
/// (`akp_merkle_tree` stores its leaf / two-to-one handles, `akp_sponge` its parameters) for their whole lifetime.  The
/// library PINS a parameter handle for every tree and sponge built on it (ABI version 3, `capi_internal.hpp` `pins`):
/// `akp_*_params_destroy` on a pinned handle only marks it and the tables are released by the last
/// `akp_merkle_tree_destroy` / `akp_sponge_destroy`.  An evicted handle that a live tree still uses therefore stays valid for
/// that tree (no use-after-free from safe Rust, whatever the eviction order), and a later `get_or_create` of the same
/// parameter set simply builds a fresh handle.  The same accounting covers the context: trees and sponges count as handles
/// of their `akp_ctx`, so a tree moved to another thread survives the exit of the thread that built it (its calls then
/// fail with a clean error instead of touching a freed context).
pub struct HandleCache<H: Copy> {
    entries: Vec<CacheEntry<H>>,
    tick: u64,
    cap: usize,
    destroy: unsafe extern "C" fn(H),
}
struct CacheEntry<H> {
    fp: u64,
    tag: u64,
    key: Vec<Fr>,
    handle: H,
    last_use: u64,
}
impl<H: Copy> HandleCache<H> {
    pub fn new(cap: usize, destroy: unsafe extern "C" fn(H)) -> Self {
        Self { entries: Vec::new(), tick: 0, cap, destroy }
    }
    /// the handle of (`tag`, `key`), created with `create` on a miss; `tag` carries the scalar parameters (dimensions, kind)
    pub fn get_or_create(&mut self, tag: u64, key: &[Fr], create: impl FnOnce() -> Result<H, Error>) -> Result<H, Error> {
        self.tick += 1;
        let fp = fingerprint(tag, key.iter());
        if let Some(e) = self.entries.iter_mut().find(|e| e.fp == fp && e.tag == tag && e.key.as_slice() == key) {
            e.last_use = self.tick;
            return Ok(e.handle);
        }
        let handle = create()?;
        if self.entries.len() >= self.cap {
            let (i, _) = self.entries.iter().enumerate().min_by_key(|(_, e)| e.last_use).expect("cap > 0");
            let old = self.entries.swap_remove(i);
            unsafe { (self.destroy)(old.handle) };
        }
        self.entries.push(CacheEntry { fp, tag, key: key.to_vec(), handle, last_use: self.tick });
        Ok(handle)
    }
}
impl<H: Copy> Drop for HandleCache<H> {
    fn drop(&mut self) {
        for e in self.entries.drain(..) {
            unsafe { (self.destroy)(e.handle) };
        }
    }
}

/// One context per OS thread + the parameter handles created on it.  A handle returned by a cache is used by the trait
/// implementations within the call that asked for it, or handed to a tree / sponge constructor, which pins it (above).
pub struct ThreadRuntime {
    // field order = drop order: the handle caches go before the context they were created on
    pub poseidon: HandleCache<*mut ffi::AkpPoseidon>,
    pub te: HandleCache<*mut ffi::AkpTeParams>,
    pub ctx: CtxGuard,
}
/// owns the thread's `akp_ctx`
pub struct CtxGuard(pub *mut ffi::AkpCtx);
impl Drop for CtxGuard {
    fn drop(&mut self) {
        unsafe { ffi::akp_ctx_destroy(self.0) };
    }
}
impl ThreadRuntime {
    fn new() -> Self {
        layout_check();
        let dev: i32 = std::env::var("AKP_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        let mut ctx = core::ptr::null_mut();
        let rc = unsafe { ffi::akp_ctx_create(dev, &mut ctx) };
        assert_eq!(rc, ffi::AKP_OK, "akp_ctx_create({dev}) failed: there is no CPU fallback");
        // Poseidon constants are a few KB per set.  The precomputed tables of a curve-hash set live ONCE per device, shared by the
        // handles of every thread's context (libakp's table store: the analogue of the reference's `&Parameters` borrowed by all
        // rayon workers, crh/mod.rs:22): this thread's handle attaches to them.  Their size follows the context's table budget --
        // by default 320 MiB (cache-sized tables, built in milliseconds); AKP_TABLE_BUDGET_MB here (or `set_table_budget` before
        // the first curve hash) raises it, `AKP_TABLE_BUDGET_MB=device` asks for the HBM-sized tables (46 GB for a 4x256 Pedersen
        // window: -23 % per hash).  Since ABI version 5 a budget above the default costs nothing at the start: the handle hashes on the
        // cache-sized table while a thread of libakp builds the wide one in the background (`akp_te_params_table_info` reports the state;
        // `akp_te_params_prepare` waits for it).
        if let Ok(v) = std::env::var("AKP_TABLE_BUDGET_MB") {
            let bytes = if v == "device" { Some(usize::MAX) } else { v.parse::<usize>().ok().map(|mb| mb << 20) };
            if let Some(b) = bytes {
                let rc = unsafe { ffi::akp_ctx_set_table_budget(ctx, b) };
                assert_eq!(rc, ffi::AKP_OK);
            }
        }
        Self { poseidon: HandleCache::new(64, ffi::akp_poseidon_params_destroy), te: HandleCache::new(4, ffi::akp_te_params_destroy), ctx: CtxGuard(ctx) }
    }
}
thread_local! {
    static RT: RefCell<Option<ThreadRuntime>> = const { RefCell::new(None) };
}
/// HBM one precomputed Pedersen / Bowe-Hopwood table may take on this thread's device (`akp_ctx_set_table_budget`; 0 = the
/// library's default).  Handles that exist keep their tables; the next `setup` / first use of new generators follows the budget.
pub fn set_table_budget(bytes: usize) -> Result<(), Error> {
    with_runtime(|rt| check(unsafe { ffi::akp_ctx_set_table_budget(rt.ctx.0, bytes) }, 0))
}
/// run `f` with this thread's runtime (created on first use)
pub fn with_runtime<R>(f: impl FnOnce(&mut ThreadRuntime) -> R) -> R {
    RT.with(|cell| {
        let mut slot = cell.borrow_mut();
        f(slot.get_or_insert_with(ThreadRuntime::new))
    })
}

/// RAII registration of a caller-owned buffer with the GPU runtime (`akp_host_register`): while the guard lives, the batch
/// entry points address the buffer directly over PCIe instead of copying it (zero copy; 3.4e8 instead of 2.9e8
/// permutations/s PCIe-inclusive on MI355X).  Registering costs more than one batch, so keep the guard for as long as
/// the buffer is reused.  The whole slice is registered: the runtime rejects copies that straddle registered and
/// unregistered memory.
pub struct Registered<'a, T> {
    buf: &'a mut [T],
}
impl<'a, T> Registered<'a, T> {
    pub fn new(buf: &'a mut [T]) -> Result<Self, Error> {
        let bytes = core::mem::size_of_val(buf);
        check(unsafe { ffi::akp_host_register(buf.as_mut_ptr() as *mut core::ffi::c_void, bytes) }, 0)?;
        Ok(Self { buf })
    }
    pub fn as_slice(&self) -> &[T] {
        self.buf
    }
    pub fn as_mut_slice(&mut self) -> &mut [T] {
        self.buf
    }
}
impl<T> Drop for Registered<'_, T> {
    fn drop(&mut self) {
        unsafe {
            ffi::akp_host_unregister(self.buf.as_mut_ptr() as *mut core::ffi::c_void);
        }
    }
}

/// FNV-1a over the limbs of field elements (cheap fingerprint for the handle caches)
pub fn fingerprint<'a>(seed: u64, elems: impl Iterator<Item = &'a Fr>) -> u64 {
    let mut h = 0xcbf2_9ce4_8422_2325u64 ^ seed;
    for e in elems {
        for w in (e.0).0 {
            h = (h ^ w).wrapping_mul(0x0000_0100_0000_01b3);
        }
    }
    h
}

/// flatten `Vec<Vec<Fr>>` (ark, mds) row-major
pub fn flatten(m: &[Vec<Fr>]) -> Vec<Fr> {
    m.iter().flat_map(|r| r.iter().copied()).collect()
}
#[allow(dead_code)]
fn _assert_prime_field<F: PrimeField>() {}
