"""The one JSON line of a run, the watchdog that protects it, and the wrapper every optional leg runs under."""
import json
import os
import time


class ResultLine:
    """The one JSON line of a run and the watchdog that protects it.  The line exists (`out`) as soon as the headline is measured;
    the optional legs that follow only add fields.  With the watchdog started, a leg that exceeds the budget it was armed with — a
    collective that never completes on an N > 1 run — costs that leg, not the line: rank 0 prints what it has, with
    `aborted_optional_leg` naming the leg, and every rank leaves with exit code 0 (os._exit: the hung thread cannot be joined)."""

    def __init__(self, fd, rank, out):
        import threading
        self.fd, self.rank, self.out = fd, rank, out
        self._emitted = threading.Event()
        self._lock = threading.Lock()
        self._leg = (None, None)                 # (name, deadline on time.monotonic())

    @staticmethod
    def _scrub(x):
        """an emulated dry run carries no timing of anything: drop every clock-derived field, keep the verdicts"""
        if isinstance(x, dict):
            return {k: ResultLine._scrub(v) for k, v in x.items()
                    if not (k == "ms" or k.endswith("_ms") or k.startswith("ms_") or "_ms_" in k or "constraints_per_s" in k or k.startswith("proof_ms"))}
        if isinstance(x, list):
            return [ResultLine._scrub(v) for v in x]
        return x

    def emit(self):
        with self._lock:
            if self.rank == 0 and not self._emitted.is_set():
                line = self._scrub(self.out) if isinstance(self.out, dict) and self.out.get("emulated") else self.out
                os.write(self.fd, (json.dumps(line) + "\n").encode())
            self._emitted.set()

    def arm(self, name, seconds):
        """Start (or, with name None, stop) the watchdog clock of one optional leg."""
        self._leg = (name, time.monotonic() + seconds) if name else (None, None)

    def start_watchdog(self, poll_s=1.0):
        import threading

        def run():
            while True:                              # daemon thread: ends with the process
                time.sleep(poll_s)
                name, deadline = self._leg
                if deadline is not None and time.monotonic() > deadline:
                    if self.out is None and name == "headline":
                        # the timed region itself never finished (a collective that hangs on an N > 1 run): there is no measurement to
                        # print — say so on rank 0 and leave with a failure code instead of hanging until the driver's own limit
                        if self.rank == 0:
                            os.write(self.fd, (json.dumps({"error": "the warm-up / timed steps exceeded the headline watchdog budget "
                                                                    "(a collective that never completed?); nothing was measured"}) + "\n").encode())
                        os._exit(4)
                    if self.rank == 0:
                        self.out["aborted_optional_leg"] = {"leg": name, "note": "the leg exceeded its watchdog budget (a collective that never "
                                                            "completed?); the headline above was measured before it started and is unaffected"}
                    self.emit()
                    os._exit(0)

        threading.Thread(target=run, daemon=True).start()


def run_leg(guard, name, budget_s, fn, error=lambda ex: {"error": repr(ex)}):
    """One optional leg of a run, after the headline: under the watchdog (`budget_s`; None = unguarded) and inside its own try/except —
    an exception costs this leg's fields (`{"error": ...}`), never the line."""
    if budget_s is not None:
        guard.arm(name, budget_s)
    try:
        return fn()
    except Exception as ex:     # noqa: BLE001 - see above
        return error(ex)
    finally:
        if budget_s is not None:
            guard.arm(None, 0)
