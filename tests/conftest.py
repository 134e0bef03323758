import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("LANDIFF_SKIP_INIT", "1")      # importing `landiff` would otherwise look for / download the released checkpoints
os.environ.setdefault("LD_TUNING", "1")              # the library re-reads its LD_* knobs on every call (tests alternate kernel forms in one process)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A test that hangs (a wedged GPU queue, a lost rendezvous) must fail, not stall the whole run: with pytest-timeout present
    # and no --timeout given, every test gets 30 minutes (the slowest one, the full-size VAE against the fp32 oracle on the host cores, takes ~4 on an idle host).
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 1800.0


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (these tests must not silently skip)")
    return torch.device("cuda:0")


class OracleJobs:
    """The slow oracle legs of the full-size GPU tests (tests/oracle_jobs.py) as child processes on the host cores: started right
    after collection for the selected tests, joined by the tests themselves -- the GPU box has 256 hardware threads that the GPU
    tests in between leave idle, and the oracle never touches the GPU.  A job nobody started yet is started by the first result()
    call (a single test run on its own)."""

    def __init__(self):
        self.dir = tempfile.mkdtemp(prefix="ld_oracle_jobs_")
        self.procs = {}
        self.parent_cpus, self.parent_threads = None, None   # what to give back to pytest once the last job has been joined
        self.plan = {}                                        # name -> cores, for jobs that a test starts later (LATE_JOBS)

    def start(self, name, cpus=None, arg=None):
        """cpus: the host cores this child (and its torch threads) may use -- the session hands every job its own share of the
        upper half of the cpuset and keeps the lower half for pytest itself, so that the oracle's OpenMP teams and the tests'
        own CPU work do not fight over cores (oversubscribed spin-waiting teams made one test 5x slower)."""
        if name in self.procs:
            return
        cpus = cpus or self.plan.get(name)
        out = os.path.join(self.dir, name + ".pt")
        log = open(os.path.join(self.dir, name + ".log"), "w")
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")     # CPU only: the checker never sees the GPU
        cmd = [sys.executable, os.path.join(ROOT, "tests", "oracle_jobs.py"), name, out] + ([arg] if arg else [])
        pre = None
        if cpus:
            env["OMP_NUM_THREADS"] = str(len(cpus)); env["LD_ORACLE_JOB_THREADS"] = str(len(cpus))
            pre = lambda: os.sched_setaffinity(0, cpus)
        p = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT, preexec_fn=pre)
        self.procs[name] = (p, out, log)

    def result(self, name, timeout=1700.0):
        import torch
        from oracle_jobs import LATE_JOBS
        assert name in self.procs or name not in LATE_JOBS, f"oracle job {name} is started by {LATE_JOBS.get(name)}: select that test too"
        self.start(name)
        p, out, log = self.procs[name]
        rc = p.wait(timeout=timeout)
        log.close()
        if rc != 0 or not os.path.exists(out):
            with open(os.path.join(self.dir, name + ".log")) as f:
                pytest.fail(f"oracle job {name} failed (exit code {rc}):\n{f.read()[-4000:]}")
        d = torch.load(out, weights_only=False)
        if self.parent_cpus and all(q.poll() is not None for q, _, _ in self.procs.values()):
            os.sched_setaffinity(0, self.parent_cpus)      # every job is done: pytest gets the whole cpuset back
            torch.set_num_threads(self.parent_threads)
            self.parent_cpus = None
        return d["result"], d["seconds"]

    def close(self):
        for p, _, log in self.procs.values():
            if p.poll() is None:
                p.kill()
                p.wait()
            if not log.closed:
                log.close()
        import shutil
        shutil.rmtree(self.dir, ignore_errors=True)


_JOBS = None


# CPU-heavy tests whose oracle legs depend on device outputs (they cannot be started ahead): run them LAST, when the background jobs
# have been joined and pytest has the whole cpuset again
RUN_LAST = ("test_llm_full_size_prefill_and_decode_vs_oracle",)
# ... and the one whose oracle leg is long enough to be worth a child process of its own: it runs FIRST, hands its latent to the
# child (LATE_JOBS) and a second test joins the result at the very end
RUN_FIRST = ("test_infer_video_entry_point_config0",)


def pytest_collection_modifyitems(session, config, items):
    """Order: everything else, then the tests that JOIN a background oracle job (by then the jobs have had the whole session to
    finish: no waiting), then the CPU-heavy tests that need the whole cpuset."""
    from oracle_jobs import CONSUMERS
    name = lambda it: it.nodeid.split("::")[-1].split("[")[0]
    join = [it for it in items if name(it) in CONSUMERS]
    last = [it for it in items if name(it) in RUN_LAST]
    first = [it for it in items if name(it) in RUN_FIRST]
    if join or last or first:
        items[:] = first + [it for it in items if it not in join and it not in last and it not in first] + join + last


def pytest_collection_finish(session):
    """Start the oracle children of the selected tests now, so that they run under the GPU tests that come first."""
    global _JOBS
    from oracle_jobs import CONSUMERS
    want = []
    for item in session.items:
        for test, jobs in CONSUMERS.items():
            if item.nodeid.endswith("::" + test):
                want += jobs
    if want and not session.config.option.collectonly:
        _JOBS = OracleJobs()
        # core plan: pytest keeps the lower half of its cpuset, the jobs split the upper half by weight (the VAE decode is the long one)
        weight = {"vae_two_chunks": 4, "llm_two_blocks_fp32": 2, "llm_two_blocks_bf16": 2, "dit_3p3_eps": 2, "dit_layer": 1, "vae_level0": 2,
                  "config0_frames": 3}
        cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
        plan = {}
        if len(cpus) >= 96:      # (a small host: no partition -- the jobs and pytest share the cores as the in-process legs used to)
            mine, theirs = cpus[: len(cpus) // 2], cpus[len(cpus) // 2:]
            tot, at = sum(weight.get(n, 1) for n in want), 0
            for n in want:
                share = max(2, len(theirs) * weight.get(n, 1) // tot)
                plan[n] = set(theirs[at:at + share]) or None
                at += share
            import torch
            _JOBS.parent_cpus, _JOBS.parent_threads = set(cpus), torch.get_num_threads()
            os.sched_setaffinity(0, set(mine))
            torch.set_num_threads(max(1, min(len(mine), 64)))
        from oracle_jobs import LATE_JOBS
        _JOBS.plan = plan
        for name in want:
            if name not in LATE_JOBS:                  # (those are started by their producer test, with its data)
                _JOBS.start(name, plan.get(name))


def pytest_sessionfinish(session, exitstatus):
    global _JOBS
    if _JOBS is not None:
        _JOBS.close()
        _JOBS = None


@pytest.fixture(scope="session")
def oracle_bg():
    global _JOBS
    if _JOBS is None:
        _JOBS = OracleJobs()
    return _JOBS
