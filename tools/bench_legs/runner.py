"""A side leg must not be able to take the headline down: every leg of bench.py runs through `LegRunner.run`, which turns anything the
leg raises (an out-of-memory on a wide table, an RCCL error, a `SystemExit` of a leg's own parity check) into a string under
`errors[name]` and lets the run go on; the line is always printed with the headline, and names the legs that failed.

At N > 1 a leg that fails on ONE rank must not leave the others waiting in a collective.  All collectives the legs use (`barrier`,
`max_over_ranks`) are ONE all-reduce(MAX) of the same two numbers [value, abort flag]:

* a rank whose leg raised sends [0, 1] once -- it pairs with whatever the other ranks call next;
* a rank that sees the flag inside `barrier` / `max_over_ranks` raises `LegAborted` there and leaves the leg; it does not answer again
  (its collective count already equals the failed rank's);
* ranks that finish a leg normally meet in one more all-reduce (`_agree`): a rank that failed after the leg's last collective is seen
  there.

So every rank leaves every leg after the same number of all-reduces, whichever rank failed where.  (Collectives a leg issues itself --
the all-gather of sub-roots inside `build_sharded` -- are outside this scheme: a rank that dies between them is what the process
group's timeout is for.)

Test hook (tests/test_bench_legs_cpu.py, tests/test_gpu_bench_contract.py): AKP_BENCH_FAIL_LEG="name[@rank][:end]" makes that leg
raise on that rank (every rank without `@rank`), before the leg runs or (`:end`) after it.  Never set by the driver."""
import os


class LegAborted(RuntimeError):
    """another rank failed inside the current leg"""


class InjectedLegFailure(RuntimeError):
    pass


class LegRunner:
    def __init__(self, torch, dist, rank, world, device_for_collectives, sync_device=None):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.cdev = device_for_collectives  # where the two numbers of a collective live ("cpu" over gloo, the GPU over nccl)
        self.sync_device = sync_device or (lambda: None)
        self.errors = {}   # leg name -> what it raised (this rank, or "another rank")
        self.in_leg = None
        fail = os.environ.get("AKP_BENCH_FAIL_LEG", "")
        self._fail_when = "end" if fail.endswith(":end") else "start"
        fail = fail[:-4] if fail.endswith(":end") else fail
        self._fail_leg, _, r = fail.partition("@")
        self._fail_rank = int(r) if r else None

    # ---- the one collective ---------------------------------------------------------------------------------------------------
    def _exchange(self, value, flag):
        if not self.dist:
            return value, flag
        t = self.torch.tensor([float(value), float(flag)], dtype=self.torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        v = t.tolist()
        return v[0], v[1]

    def _collective(self, value):
        v, flag = self._exchange(value, 0.0)
        if flag:
            raise LegAborted("another rank failed in leg %r" % (self.in_leg,))
        return v

    def barrier(self):
        self.sync_device()
        self._collective(0.0)
        self.sync_device()

    def max_over_ranks(self, x):
        return self._collective(x) if self.dist else x

    # ---- running a leg --------------------------------------------------------------------------------------------------------
    def _inject(self, name, when):
        if name == self._fail_leg and when == self._fail_when and self._fail_rank in (None, self.rank):
            raise InjectedLegFailure("AKP_BENCH_FAIL_LEG: injected failure of leg %r on rank %d (%s)" % (name, self.rank, when))

    def run(self, name, fn, *args, cleanup=None, **kw):
        """fn(*args, **kw) or None; never raises (KeyboardInterrupt excepted)"""
        self.in_leg = name
        out, answered = None, False
        try:
            self._inject(name, "start")
            out = fn(*args, **kw)
            self._inject(name, "end")
        except KeyboardInterrupt:
            raise
        except LegAborted as exc:
            out, answered = None, True  # this rank's last collective WAS the failed rank's flag
            self.errors[name] = str(exc)
        except BaseException as exc:  # noqa: BLE001 -- SystemExit of a leg's parity check included
            out = None
            self.errors[name] = ("%s: %s" % (type(exc).__name__, exc))[:400]
            try:
                self._exchange(0.0, 1.0)
            except BaseException as exc2:  # noqa: BLE001 -- the process group itself is gone: nothing more to agree on
                self.errors[name] += " (and the abort flag could not be sent: %r)" % (exc2,)
            answered = True
        finally:
            self.in_leg = None
            if cleanup:
                try:
                    cleanup()
                except BaseException:  # noqa: BLE001
                    pass
        if not answered:
            _, flag = self._exchange(0.0, 0.0)
            if flag:  # this rank's result stands; the leg is still reported as failed
                self.errors[name] = "another rank failed in leg %r after its last collective" % (name,)
        return out
