"""Long single-threaded CPU computations (the checker's side of the full-size
parity tests) run in background threads while the GPU side of the tests goes
on, several at a time, under a budget of host memory.

A job is a callable whose heavy part releases the GIL (a ctypes call into the
CPU restatement); it is given the number of gigabytes it holds while it runs,
and jobs start in submission order as the budget allows.  `result(name)` waits
for one of them and, while waiting, reports progress to the uncaptured stdout
and to gpurun_out/heartbeat.log, so that a long silent wait does not look like
a hang.
"""
import os
import threading
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def host_memory_gb():
    """Physical memory of the machine and what is available now, in GB."""
    total = avail = 0.0
    with open("/proc/meminfo") as f:
        for line in f:
            if line.startswith("MemTotal:"):
                total = float(line.split()[1]) / 1e6
            elif line.startswith("MemAvailable:"):
                avail = float(line.split()[1]) / 1e6
    return total, avail


class CpuJobs:
    def __init__(self, budget_gb):
        self.budget = float(budget_gb)
        self.in_use = 0.0
        self.cv = threading.Condition()
        self.jobs = {}      # name -> dict(fn, gb, state, value, error, t0, t1)
        self.order = []
        self.t_start = time.time()

    def submit(self, name, fn, gb):
        with self.cv:
            assert name not in self.jobs
            self.jobs[name] = {"fn": fn, "gb": min(float(gb), self.budget), "state": "queued", "value": None,
                               "error": None, "t0": None, "t1": None}
            self.order.append(name)
        th = threading.Thread(target=self._run, args=(name,), daemon=True)
        th.start()

    def _run(self, name):
        job = self.jobs[name]
        with self.cv:
            # strictly in submission order: a small job never overtakes a large one that is waiting for room
            while not (self._next_queued() == name and self.in_use + job["gb"] <= self.budget + 1e-9):
                self.cv.wait()
            self.in_use += job["gb"]
            job["state"] = "running"
            job["t0"] = time.time()
            self.cv.notify_all()
        try:
            job["value"] = job["fn"]()
        except BaseException as e:  # reported by result()
            job["error"] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
        finally:
            with self.cv:
                job["t1"] = time.time()
                job["state"] = "done"
                job["fn"] = None
                self.in_use -= job["gb"]
                self.cv.notify_all()

    def _next_queued(self):
        for n in self.order:
            if self.jobs[n]["state"] == "queued":
                return n
        return None

    def status(self):
        now = time.time()
        parts = []
        for n in self.order:
            j = self.jobs[n]
            if j["state"] == "running":
                parts.append("%s running %.0f s" % (n, now - j["t0"]))
            elif j["state"] == "queued":
                parts.append("%s queued" % n)
        return ", ".join(parts) if parts else "all CPU jobs done"

    def result(self, name, capfd=None, every=20.0):
        """Value of job `name` (waits for it).  Returns (value, seconds it ran)."""
        job = self.jobs[name]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join(ROOT, "gpurun_out", "heartbeat.log")

        def wait():
            last = time.time()
            with self.cv:
                while job["state"] != "done":
                    self.cv.wait(timeout=1.0)
                    if time.time() - last >= every and job["state"] != "done":
                        last = time.time()
                        line = "[waiting for %s; %s]" % (name, self.status())
                        with open(path, "a") as f:
                            f.write(line + "\n")
                        print(line, flush=True)

        if capfd is not None:
            with capfd.disabled():
                wait()
        else:
            wait()
        if job["error"]:
            raise RuntimeError("CPU job %s failed:\n%s" % (name, job["error"]))
        return job["value"], job["t1"] - job["t0"]
