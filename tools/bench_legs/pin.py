"""Pin status of the curve half of the oracle, and the one thing that costs nothing: if the box this runs on HAS a Rust toolchain
(`cargo` on PATH) with the arkworks crates resolvable offline, run shim/examples/emit_vectors.rs once -- it executes the REFERENCE
crates on tests/golden/emitter_inputs.json and writes tests/golden/reference_vectors.json -- so that the pinning tests turn on.
Nothing of the reference is vendored or copied to make this happen; without cargo the status says so in one key."""
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VECTORS = os.path.join(ROOT, "tests", "golden", "reference_vectors.json")


def try_emit(timeout_s=600):
    """returns a one-line note on what was attempted"""
    if os.path.exists(VECTORS):
        return "present"
    cargo = shutil.which("cargo")
    if not cargo:
        return "cargo absent on this box (no Rust toolchain): emitter not run"
    env = dict(os.environ, AKP_LIB_DIR=os.path.join(ROOT, "crypto_primitives_amd", "lib"), CARGO_NET_OFFLINE="true")
    try:
        p = subprocess.run([cargo, "run", "--offline", "--release", "--example", "emit_vectors"], cwd=os.path.join(ROOT, "shim"), env=env,
                           capture_output=True, text=True, timeout=timeout_s)
    except Exception as exc:  # noqa: BLE001
        return "cargo present, emitter did not finish: %r" % (exc,)
    if p.returncode != 0 or not os.path.exists(VECTORS):
        tail = (p.stderr or p.stdout or "").strip().splitlines()[-1:] or [""]
        return "cargo present, offline build of the emitter failed (crates not in the local registry?): %s" % tail[0][:160]
    return "emitted by cargo on this box"


def status(attempt=True):
    """(short status string, note).  short status starts with 'pinned' or 'unpinned'"""
    note = try_emit() if attempt else ("present" if os.path.exists(VECTORS) else
                                       "not attempted (AKP_RUN_EMITTER=1 runs shim/examples/emit_vectors.rs; cargo %s on this box)" % ("present" if shutil.which("cargo") else "absent"))
    if not os.path.exists(VECTORS):
        return "unpinned (emitter not run): tests/golden/reference_vectors.json is absent -- shim/examples/README.md", note
    try:
        meta = json.load(open(VECTORS)).get("emitter", {})
        return "pinned by tests/golden/reference_vectors.json (emitter %s, Cargo.lock %s)" % (meta.get("git_sha", "?"), meta.get("cargo_lock_sha256", "?")[:16]), note
    except Exception as exc:  # noqa: BLE001
        return "reference_vectors.json present but unreadable: %r" % (exc,), note
