"""Single-node launcher: the Kubernetes replacement (new layer N4, externals X1/X7).

The reference hands its manifests to a kube-apiserver through the python
``kubernetes`` client (``CoreV1Api.create_namespaced_service`` /
``BatchV1Api.create_namespaced_job`` ``handlers/train_handler.py:52,112``;
``delete_namespaced_job/service`` ``service/remove_ps.py:31-33,49``; watch stream
``service/tf_cleaner.py:40``).  :class:`LocalCluster` exposes those same calls on
one node:

* a *Service* becomes an entry in a loopback "cluster DNS" (``tfId:2222`` ->
  ``127.0.0.1:<port>``) that task processes receive through ``DTF_DNS``;
* a *Job* becomes one OS process (the container "image" is a python module,
  the container command is forwarded unchanged as argv) pinned to the
  ``amd.com/gpu`` devices it requested through ``HIP_VISIBLE_DEVICES``; a Job
  that does not fit stays Pending until devices are released, like an
  unschedulable pod;
* the Job controller's status reporting becomes ADDED / MODIFIED / DELETED
  events with ``status.active/succeeded/failed``, replayable from a
  ``resourceVersion`` (the reference's watch has no resume);
* Foreground delete becomes SIGTERM -> grace -> SIGKILL on the task's process
  group (ranks blocked in an RCCL collective ignore SIGTERM) followed by an
  HBM-reclaim check of the devices the task held.
"""
from __future__ import annotations

import copy
import glob
import json
import logging
import os
import signal
import subprocess
import sys
import threading
import time

from ..utils.config import ApiConfig

log = logging.getLogger(__name__)


class ApiException(Exception):
    def __init__(self, status=500, reason=""):
        super().__init__("(%s) %s" % (status, reason))
        self.status, self.reason = status, reason


class AttrDict(dict):
    """dict with attribute access, so both ``obj["metadata"]["name"]`` and the
    reference's ``event['object'].metadata.name`` / ``status.succeeded`` work."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            return None
        return v

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return AttrDict({k: AttrDict.wrap(v) for k, v in o.items()})
        if isinstance(o, list):
            return [AttrDict.wrap(v) for v in o]
        return o


# ---------------------------------------------------------------------------
# HBM accounting without touching HIP in the service process
# ---------------------------------------------------------------------------

def _hbm_sysfs():
    cards = []
    for p in glob.glob("/sys/class/drm/card[0-9]*/device/mem_info_vram_used"):
        try:
            idx = int(p.split("/card")[1].split("/")[0])
        except ValueError:
            continue
        cards.append((idx, p))
    out = []
    for _, p in sorted(cards):
        try:
            with open(p) as f:
                out.append(int(f.read().strip()))
        except (OSError, ValueError):
            out.append(-1)
    return out


def _hbm_rocm_smi():
    import shutil

    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if not os.path.exists(exe):
        return []
    try:
        r = subprocess.run([exe, "--showmeminfo", "vram", "--json"], capture_output=True, text=True, timeout=20)
        txt = r.stdout[r.stdout.index("{"):]
        doc = json.loads(txt)
    except (OSError, ValueError, subprocess.SubprocessError):
        return []
    cards = []
    for k, v in doc.items():
        if k.startswith("card"):
            try:
                cards.append((int(k[4:]), int(v.get("VRAM Total Used Memory (B)", -1))))
            except (TypeError, ValueError):
                pass
    return [u for _, u in sorted(cards)]


_hbm_source = None


def hbm_used_bytes():
    """Per-device used VRAM in bytes without touching HIP in the service process: sysfs
    (``mem_info_vram_used``) when the container exposes it, else ``rocm-smi --json``.
    Empty list when no amdgpu device is visible (CPU box)."""
    global _hbm_source
    if _hbm_source in (None, "sysfs"):
        out = _hbm_sysfs()
        if out:
            _hbm_source = "sysfs"
            return out
    if _hbm_source in (None, "rocm-smi"):
        out = _hbm_rocm_smi()
        if out:
            _hbm_source = "rocm-smi"
            return out
    if _hbm_source is None:
        _hbm_source = "none"
    return []


def probe_hbm(gpus, python=None, timeout=90):
    """``[{"gpu", "free", "total", "used"}]`` measured by a fresh child process through
    hipMemGetInfo (the service process itself never initialises HIP)."""
    code = ("import json,torch;out=[]\n"
            "for i in range(torch.cuda.device_count()):\n"
            "    f,t=torch.cuda.mem_get_info(i);out.append({'free':f,'total':t,'used':t-f})\n"
            "print('HBM'+json.dumps(out))")
    env = dict(os.environ, HIP_VISIBLE_DEVICES=",".join(str(g) for g in gpus))
    try:
        r = subprocess.run([python or sys.executable, "-c", code], env=env, capture_output=True, text=True,
                           timeout=timeout)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("HBM")][-1]
        out = json.loads(line[3:])
    except (OSError, IndexError, ValueError, subprocess.SubprocessError):
        return None
    for g, o in zip(gpus, out):
        o["gpu"] = g
    return out


def visible_gpu_count(default=0):
    n = len(hbm_used_bytes())
    if n:
        return n
    if os.path.exists("/dev/kfd"):
        try:
            import torch

            return int(torch.cuda.device_count())      # does not initialise the GPU
        except Exception:
            pass
    return default


class _Task:
    def __init__(self, name, namespace, body):
        self.name, self.namespace, self.body = name, namespace, body
        self.proc = None
        self.gpus = []
        self.requested_gpus = 0
        self.status = {"active": 0, "succeeded": None, "failed": None,
                       "startTime": None, "completionTime": None, "phase": "Pending"}
        self.deleted = False
        self.released = False      # its GPUs went back to the free list (exactly once)
        self.gang, self.gang_gpus = None, 0     # job uid and the devices ALL its tasks need together (manifest.gen_job)
        self.log_path = None
        self.exit_code = None
        self.hbm_before = None     # sysfs / rocm-smi counters at start (when the container exposes them)
        self.hbm_during = None     # one hipMemGetInfo sample while the task runs


ENTRY_IMAGE = "distributed_tensorflow_amd.service.entry"


class _WarmPool:
    """A few task processes started ahead of time (``service/warm_worker.py``): interpreter up, torch and this
    package imported, HIP untouched.  ``take()`` hands one out (or None when none is ready -- the caller then
    starts a cold process); the pool refills in the background."""

    def __init__(self, size, python, env, root):
        self.size, self._python, self._env, self._root = int(size), python, dict(env), root
        self._refill_delay = 3.0
        self._ready = []
        self._starting = 0
        self._lock = threading.Lock()
        self._closed = False
        for _ in range(self.size):
            self._spawn_async()

    def _spawn_async(self):
        with self._lock:
            if self._closed or len(self._ready) + self._starting >= self.size:
                return
            self._starting += 1
        threading.Thread(target=self._spawn, name="warm-spawn", daemon=True).start()

    def _spawn(self):
        p = None
        try:
            env = dict(self._env)
            env["PYTHONPATH"] = self._root + os.pathsep + env.get("PYTHONPATH", "")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            p = subprocess.Popen([self._python, "-u", "-m", "distributed_tensorflow_amd.service.warm_worker"], env=env,
                                 stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 start_new_session=True)
            line = p.stdout.readline()          # "WARM <pid>": imports done
            if not line.startswith(b"WARM"):
                raise RuntimeError("warm worker did not come up")
        except Exception:       # noqa: BLE001 -- a pool that cannot fill just means cold starts
            log.exception("warm worker failed to start")
            if p is not None:
                p.kill()
            p = None
        with self._lock:
            self._starting -= 1
            if p is not None and not self._closed:
                self._ready.append(p)
                p = None
        if p is not None:
            p.kill()

    def take(self):
        with self._lock:
            while self._ready:
                p = self._ready.pop(0)
                if p.poll() is None:
                    break
                p = None
            else:
                p = None
        # refill, but not while the task that was just handed out is starting (imports are CPU-heavy)
        for _ in range(2 if p is not None else 1):
            timer = threading.Timer(self._refill_delay, self._spawn_async)
            timer.daemon = True
            timer.start()
        return p

    def ready(self):
        with self._lock:
            return len(self._ready)

    def close(self):
        with self._lock:
            self._closed = True
            procs, self._ready = self._ready, []
        for p in procs:
            try:
                p.stdin.close()
                p.kill()
            except OSError:
                pass


class _HotPool:
    """One GPU-HOT warm rank per free device (``service/warm_worker.py`` with ``DTF_WARM_HOT=1``): started with
    ``HIP_VISIBLE_DEVICES`` already pinned to that device, HIP context up, kernel library / MIOpen / hipBLASLt
    loaded.  ``take(g)`` hands out the worker of device ``g`` (None when it is not ready: the caller then starts the
    task another way, and a worker still coming up on that device is stopped so the task is alone on it)."""

    def __init__(self, python, env, root, warm_model=""):
        self._python, self._env, self._root, self._model = python, dict(env), root, warm_model
        self._lock = threading.Lock()
        self._procs = {}        # gpu -> [Popen or None while spawning, ready, generation]
        self._hint = ""         # warm_model = auto: what the last built-in GPU job on this node trained
        self._gen = 0
        self._closed = False

    def spawn(self, g):
        with self._lock:
            if self._closed or g in self._procs:
                return
            self._gen += 1
            self._procs[g] = [None, False, self._gen]
            gen = self._gen
        threading.Thread(target=self._spawn, args=(g, gen), name="hot-spawn-%d" % g, daemon=True).start()

    def _spawn(self, g, gen):
        p = None
        try:
            env = dict(self._env)
            env["PYTHONPATH"] = self._root + os.pathsep + env.get("PYTHONPATH", "")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.update(HIP_VISIBLE_DEVICES=str(g), DTF_WARM_HOT="1", DTF_NUM_GPUS="1")
            if self._model and self._model != "auto":
                env["DTF_WARM_MODEL"] = self._model
            elif self._model == "auto" and self._hint:
                env["DTF_WARM_MODEL"] = self._hint
            p = subprocess.Popen([self._python, "-u", "-m", "distributed_tensorflow_amd.service.warm_worker"], env=env,
                                 stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 start_new_session=True)
            with self._lock:
                cur = self._procs.get(g)
                if cur is not None and cur[2] == gen:
                    cur[0] = p
            line = p.stdout.readline()          # "WARM <pid> HOT": device initialised, libraries loaded
            if not line.startswith(b"WARM"):
                raise RuntimeError("hot worker for GPU %d did not come up" % g)
        except Exception:       # noqa: BLE001 -- no hot worker just means a colder start
            with self._lock:
                cur = self._procs.get(g)
                mine = cur is not None and cur[2] == gen
                if mine:
                    self._procs.pop(g, None)
            if mine and not self._closed:
                log.exception("hot worker for GPU %d failed to start", g)
            else:
                # take() / close() stopped it while it was coming up: a task got the device first (a burst of jobs)
                log.info("hot worker for GPU %d was stopped while starting: a task took the device", g)
            if p is not None:
                p.kill()
            return
        with self._lock:
            cur = self._procs.get(g)
            if cur is not None and cur[2] == gen and not self._closed:
                cur[1] = True
                p = None
        if p is not None:       # discarded while it was starting
            p.kill()

    def observe(self, file_field):
        """``warm_model = auto``: remember the model / batch / shape options of the job that just started; the ranks
        spawned from now on run that once while idle (a node mostly re-runs the model family it ran last)."""
        if self._model != "auto" or not str(file_field).startswith("builtin://"):
            return
        try:
            from ..train.jobspec import parse_file_field

            _, opts = parse_file_field(file_field)
            q = ["batch=%d" % opts.batch] + ["%s=%s" % kv for kv in sorted(opts.extra.items())]
            self._hint = "%s?%s" % (opts.model, "&".join(q))
        except Exception:       # noqa: BLE001 -- a hint is an optimisation
            pass

    def is_ready(self, g):
        with self._lock:
            cur = self._procs.get(g)
            return bool(cur and cur[1] and cur[0] is not None and cur[0].poll() is None)

    def ready(self):
        with self._lock:
            return sorted(g for g, cur in self._procs.items() if cur[1] and cur[0].poll() is None)

    def take(self, g):
        """The ready worker of device ``g``, or None.  Either way the device has no hot worker afterwards."""
        with self._lock:
            cur = self._procs.pop(g, None)
        if cur is None:
            return None
        p, ready, _ = cur
        if p is not None and ready and p.poll() is None:
            return p
        if p is not None:
            try:
                p.kill()
                p.wait(5)
            except (OSError, subprocess.SubprocessError):
                pass
        return None

    def discard(self, g):
        p = self.take(g)
        if p is not None:
            try:
                p.stdin.close()
                p.kill()
                p.wait(5)
            except (OSError, subprocess.SubprocessError):
                pass

    def close(self):
        with self._lock:
            self._closed = True
            gpus = list(self._procs)
        for g in gpus:
            self.discard(g)


class LocalCluster:
    def __init__(self, gpus=None, log_dir=None, python=None, extra_env=None, cfg=None):
        cfg = cfg or ApiConfig()
        self._cfg = cfg
        if gpus is None:
            n = visible_gpu_count(0)
            cap = cfg.getint("launcher", "gpus_per_node", 8)
            gpus = list(range(min(n, cap))) if n else []
        self._free_gpus = list(gpus)
        self._all_gpus = list(gpus)
        self._gang_fifo = cfg.getboolean("launcher", "gang_fifo", True)
        self._gangs = {}            # job uid -> devices set aside for its tasks that have not started yet
        self._gang_broken = set()   # jobs being torn down: no reservation any more
        self._log_dir = log_dir or cfg.get("launcher", "log_dir", "logs/tasks")
        # notes about running tasks (reap_orphans): NEXT TO the log directory, which holds nothing but task logs
        self._run_dir = os.path.abspath(self._log_dir).rstrip(os.sep) + ".running"
        self._python = python or sys.executable
        self._extra_env = dict(extra_env or {})
        self._grace = cfg.getfloat("launcher", "grace_seconds", 5)
        self._kill_wait = cfg.getfloat("launcher", "kill_seconds", 5)
        self._port_next = cfg.getint("launcher", "port_base", 23000)
        self._lock = threading.RLock()
        self._cv = threading.Condition(self._lock)
        self._services = {}   # (ns, name) -> body(with status.hostPort)
        self._jobs = {}       # (ns, name) -> _Task
        self._events = []     # (rv, type, snapshot)
        self._rv = 0
        self._closed = False
        self.reclaim_reports = []   # one dict per ended GPU task (success path included)
        self._max_events = cfg.getint("launcher", "max_events", 4096)
        self._event_ttl = cfg.getfloat("launcher", "event_ttl_seconds", 600)
        self._gone = {}             # (ns, name) of deleted jobs -> deletion time (event compaction)
        # hipMemGetInfo "used" of every device while NO task holds it: what a reclaimed device must return to
        self._idle_used = {}
        self._verify = cfg.getboolean("launcher", "verify_reclaim", True) if self._all_gpus else False
        self._sample_wait = cfg.getfloat("launcher", "hbm_sample_wait_seconds", 60.0)
        # finished Jobs are kept (the reference never deletes worker Jobs: CleanJobHandler only removes the ps side),
        # which in a long-lived service grows without bound: a finished Job is deleted ``ttlSecondsAfterFinished``
        # (Job spec, the k8s field) or ``finished_job_ttl_seconds`` (config default for all Jobs) after it ended; 0 = keep
        self._finished_ttl = cfg.getfloat("launcher", "finished_job_ttl_seconds", 0.0)
        # tasks a previous service instance left behind hold devices this instance is about to hand out
        try:
            self.orphans_reaped = self.reap_orphans() if cfg.getboolean("launcher", "reap_orphans", True) else []
        except Exception:       # noqa: BLE001 -- never a reason not to start
            log.exception("orphan sweep failed")
            self.orphans_reaped = []
        # warm rank pools: only for the framework's own entry module (a custom image starts cold)
        n_warm = cfg.getint("launcher", "warm_pool", 0)
        self._root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        self._warm = None
        own_image = cfg.get("image", "tensorflow") == ENTRY_IMAGE
        if n_warm > 0 and own_image:
            self._warm = _WarmPool(n_warm, self._python, dict(os.environ, **self._extra_env), self._root)
        # GPU-hot ranks: one per free device, started AFTER the idle-device baseline has been probed (a hot rank
        # holds a HIP context, i.e. HBM, on its device)
        self._hot = None
        if self._all_gpus and own_image and cfg.getboolean("launcher", "warm_gpu_pool", False):
            self._hot = _HotPool(self._python, dict(os.environ, **self._extra_env), self._root,
                                 cfg.get("launcher", "warm_model", ""))
        if self._verify:
            threading.Thread(target=self._probe_idle_baseline, name="hbm-baseline", daemon=True).start()
        elif self._hot is not None:
            for g in self._all_gpus:
                self._hot.spawn(g)

    # ------------------------------------------------------------------ DNS
    def _alloc_port(self):
        import socket

        for _ in range(2000):
            port = self._port_next
            self._port_next += 1
            if self._port_next > 60000:
                self._port_next = self._cfg.getint("launcher", "port_base", 23000)
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
                try:
                    s.bind(("127.0.0.1", port))
                    return port
                except OSError:
                    continue
        raise ApiException(507, "no free loopback port for service")

    def resolve(self, name, namespace=None):
        with self._lock:
            for (ns, n), body in self._services.items():
                if n == name and (namespace is None or ns == namespace):
                    return body["status"]["hostPort"]
        return None

    def dns_map(self, uid=None):
        with self._lock:
            return {n: b["status"]["hostPort"] for (_, n), b in self._services.items()
                    if uid is None or ("-" + str(uid) + "-") in n}

    # ------------------------------------------------------------- services
    def create_namespaced_service(self, namespace, body):
        name = body["metadata"]["name"]
        with self._lock:
            if (namespace, name) in self._services:
                raise ApiException(409, "service %s already exists" % name)
            b = copy.deepcopy(body)
            b.setdefault("metadata", {})["namespace"] = namespace
            b["status"] = {"hostPort": "127.0.0.1:%d" % self._alloc_port()}
            self._services[(namespace, name)] = b
            return AttrDict.wrap(b)

    def delete_namespaced_service(self, name, namespace, body=None):
        with self._lock:
            b = self._services.pop((namespace, name), None)
        if b is None:
            raise ApiException(404, "service %s not found" % name)
        return AttrDict.wrap({"status": "Success", "details": {"name": name, "kind": "services"}})

    def list_namespaced_service(self, namespace):
        with self._lock:
            return AttrDict.wrap({"items": [copy.deepcopy(b) for (ns, _), b in self._services.items()
                                            if ns == namespace]})

    # ----------------------------------------------------------------- jobs
    def _snapshot(self, t):
        b = copy.deepcopy(t.body)
        b.setdefault("metadata", {})["namespace"] = t.namespace
        b["status"] = dict(t.status)
        b["status"]["gpus"] = list(t.gpus)
        b["status"]["exitCode"] = t.exit_code
        b["status"]["logPath"] = t.log_path
        return b

    def _emit(self, typ, t):
        # caller holds the lock
        self._rv += 1
        snap = self._snapshot(t)
        snap["metadata"]["resourceVersion"] = str(self._rv)
        self._events.append((self._rv, typ, snap))
        if typ == "DELETED":
            self._gone[(t.namespace, t.name)] = time.time()
        if len(self._events) > self._max_events:
            self._compact_events()
        self._cv.notify_all()

    def _compact_events(self):
        """Bound the event log: events of jobs deleted more than ``event_ttl_seconds`` ago are dropped (every
        watcher has long seen them); if that is not enough the oldest events go.  Watchers address events by
        resourceVersion, never by position, so trimming cannot make them skip or repeat."""
        now = time.time()
        dead = {k for k, ts in self._gone.items() if now - ts > self._event_ttl}
        if dead:
            self._events = [e for e in self._events
                            if (e[2]["metadata"].get("namespace"), e[2]["metadata"]["name"]) not in dead]
            for k in dead:
                self._gone.pop(k, None)
        if len(self._events) > self._max_events:
            del self._events[:len(self._events) - self._max_events]

    def _probe_idle_baseline(self):
        """One hipMemGetInfo probe of the idle devices at service start (a child process; the service never
        initialises HIP).  A reclaimed device must come back to within ``hbm_reclaim_slack_mb`` of this."""
        with self._lock:
            gpus = [g for g in self._all_gpus if g in self._free_gpus]
        if not gpus:
            return
        out = probe_hbm(gpus, self._python)
        if out:
            with self._lock:
                for o in out:
                    # only trust a baseline measured while the device was still unassigned
                    if o["gpu"] in self._free_gpus:
                        self._idle_used.setdefault(o["gpu"], o["used"])
        if self._hot is not None:
            with self._lock:
                free = list(self._free_gpus)
            for g in free:
                self._hot.spawn(g)

    @staticmethod
    def _gpu_request(body, resource_name):
        n = 0
        for c in body["spec"]["template"]["spec"]["containers"]:
            n += int(((c.get("resources") or {}).get("limits") or {}).get(resource_name, 0))
        return n

    def create_namespaced_job(self, namespace, body):
        name = body["metadata"]["name"]
        with self._lock:
            if self._closed:
                raise ApiException(503, "cluster is shut down")
            if (namespace, name) in self._jobs:
                raise ApiException(409, "job %s already exists" % name)
            t = _Task(name, namespace, copy.deepcopy(body))
            t.requested_gpus = self._gpu_request(body, self._cfg.get("launcher", "gpu_resource", "amd.com/gpu"))
            if t.requested_gpus > len(self._all_gpus) and self._all_gpus:
                raise ApiException(422, "job %s requests %d %s, node has %d" % (
                    name, t.requested_gpus, "amd.com/gpu", len(self._all_gpus)))
            ann = (body.get("metadata") or {}).get("annotations") or {}
            try:
                t.gang, t.gang_gpus = ann.get("dtf/gang") or None, int(ann.get("dtf/gang-gpus", 0) or 0)
            except (TypeError, ValueError):
                t.gang, t.gang_gpus = None, 0
            if t.gang and t.gang_gpus > len(self._all_gpus) and self._all_gpus:
                # its tasks need these devices AT THE SAME TIME: admitted, some would start, wait in the rendezvous for
                # the ones that never can, and hold their devices until a deadline
                raise ApiException(422, "the tasks of job %s need %d %s at once, node has %d" % (
                    t.gang, t.gang_gpus, "amd.com/gpu", len(self._all_gpus)))
            self._jobs[(namespace, name)] = t
            self._emit("ADDED", t)
            self._schedule()
            return AttrDict.wrap(self._snapshot(t))

    def _schedule(self):
        # caller holds the lock; start every pending task that fits (FIFO)
        waiting = False     # an earlier job is waiting for ALL its devices: later jobs do not take devices past it
        for t in list(self._jobs.values()):
            if t.proc is not None or t.deleted or t.status["phase"] != "Pending":
                continue
            need = t.requested_gpus if self._all_gpus else 0
            if waiting and self._gang_fifo and (need > 0 or (t.gang and t.gang_gpus > 0)) and t.gang not in self._gangs:
                continue
            pool = self._free_gpus
            if t.gang and t.gang_gpus > 0 and self._all_gpus and t.gang not in self._gang_broken:
                # all-or-nothing per job: the first task that finds the WHOLE job's devices free sets them aside, the
                # job's other tasks draw from that reservation; until then every task of the job (its device-less ps
                # tasks too: they would sit in the rendezvous) stays Pending.  Two jobs submitted at the same moment
                # can therefore never hold half of the node each and wait for the other half.
                if t.gang not in self._gangs:
                    if t.gang_gpus > len(self._free_gpus):
                        # no back-filling past a waiting job: with it, a stream of small jobs keeps the devices busy and
                        # a job that needs the whole node never finds them all free
                        waiting = True
                        continue
                    if self._hot is not None:
                        self._free_gpus.sort(key=lambda g: (not self._hot.is_ready(g), g))
                    self._gangs[t.gang] = [self._free_gpus.pop(0) for _ in range(t.gang_gpus)]
                    self._free_gpus.sort()
                pool = self._gangs[t.gang]
            if need > len(pool):
                continue
            if need == 1 and self._hot is not None:
                # prefer a device whose GPU-hot rank is ready
                g = next((g for g in pool if self._hot.is_ready(g)), pool[0])
                pool.remove(g)
                t.gpus = [g]
            else:
                t.gpus = [pool.pop(0) for _ in range(need)]
            try:
                self._start(t)
            except Exception as e:  # spawn failure = failed job, not a dead service
                log.exception("failed to start %s", t.name)
                self._free_gpus.extend(t.gpus)
                self._free_gpus.sort()
                t.gpus = []
                t.status.update(active=0, failed=1, phase="Failed", completionTime=time.time())
                t.exit_code = -1
                t.status["message"] = str(e)
                self._emit("MODIFIED", t)

    def _start(self, t):
        c = t.body["spec"]["template"]["spec"]["containers"][0]
        image = c.get("image") or self._cfg.get("image", "tensorflow")
        argv = list(c.get("command") or [])
        # the container entrypoint path is symbolic: argv[1:] is the contract
        args = argv[1:] if argv and argv[0].endswith("entry.sh") else argv
        cmd = [self._python, "-u", "-m", image] + [str(a) for a in args]
        env = dict(os.environ)
        env.update(self._extra_env)
        for kv in c.get("env") or []:
            env[str(kv["name"])] = str(kv.get("value", ""))
        uid = t.name.split("-ps-")[0].split("-worker-")[0][3:]
        env["DTF_TASK_NAME"] = t.name
        env["DTF_DNS"] = json.dumps(self.dns_map(uid))
        env["DTF_NUM_GPUS"] = str(len(t.gpus))
        if t.gpus:
            env["HIP_VISIBLE_DEVICES"] = ",".join(str(g) for g in t.gpus)
        elif self._all_gpus:
            env["HIP_VISIBLE_DEVICES"] = ""    # host-side task: no device
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        os.makedirs(self._log_dir, exist_ok=True)
        t.log_path = os.path.join(self._log_dir, t.name + ".log")
        warm, hot = None, False
        if self._hot is not None and t.gpus and image == ENTRY_IMAGE and len(args) > 4:
            self._hot.observe(args[4])
        if self._hot is not None and t.gpus:
            if image == ENTRY_IMAGE and len(t.gpus) == 1:
                warm = self._hot.take(t.gpus[0])
                hot = warm is not None
            for g in t.gpus:
                self._hot.discard(g)            # the task must be alone on its devices
        if warm is None and self._warm is not None and image == ENTRY_IMAGE:
            warm = self._warm.take()
        if warm is not None:
            # a pre-started process becomes the task: it only needs what differs from the launcher's own environment
            delta = {k: v for k, v in env.items() if os.environ.get(k) != v}
            try:
                warm.stdin.write((json.dumps({"argv": [str(a) for a in args], "env": delta,
                                              "log": t.log_path}) + "\n").encode())
                warm.stdin.flush()
                warm.stdin.close()
                t.proc = warm
                t.status["warm"] = "hot" if hot else True
            except OSError:
                warm.kill()
                warm = None
        if warm is None:
            logf = open(t.log_path, "ab", buffering=0)
            try:
                t.proc = subprocess.Popen(cmd, env=env, stdout=logf, stderr=subprocess.STDOUT,
                                          stdin=subprocess.DEVNULL, start_new_session=True)
            finally:
                logf.close()
        t.status.update(active=1, phase="Running", startTime=time.time())
        self._emit("MODIFIED", t)
        self._register_running(t)
        threading.Thread(target=self._reap, args=(t,), name="wait-" + t.name, daemon=True).start()
        deadline = t.body["spec"].get("activeDeadlineSeconds")
        if deadline and not t.name.split("-")[-3] == "ps":
            timer = threading.Timer(float(deadline), self._deadline_exceeded, args=(t,))
            timer.daemon = True
            timer.start()

    def _reap(self, t):
        # counters / probes run here, in the task's own waiter thread: never under the cluster lock
        if t.gpus and self._verify:
            used = hbm_used_bytes()
            t.hbm_before = {g: used[g] for g in t.gpus if g < len(used)} or None
            # one sample while the task holds the device -- taken once its log reports the first finished step (the
            # model, optimizer state and activations are allocated by then: a sample at a fixed delay used to land
            # before the allocation and read the idle value), else after hbm_sample_wait_seconds
            t_end = time.time() + self._sample_wait
            seen = False
            while t.proc.poll() is None and time.time() < t_end and not seen:
                try:
                    with open(t.log_path, "rb") as f:
                        seen = b"first step done" in f.read()
                except OSError:
                    pass
                if not seen:
                    try:
                        t.proc.wait(max(0.01, min(0.25, t_end - time.time())))
                    except subprocess.TimeoutExpired:
                        pass
            if t.proc.poll() is None:
                t.hbm_during = probe_hbm(t.gpus, self._python)
                if t.hbm_during:
                    for d in t.hbm_during:
                        d["after_first_step"] = seen
        rc = t.proc.wait()
        self._unregister_running(t)
        self._sweep_stage(t)
        with self._lock:
            t.exit_code = rc
            if t.deleted:
                return        # delete_namespaced_job owns the rest: kill -> verify -> release, exactly once
            if rc == 0:
                t.status.update(active=0, succeeded=1, phase="Succeeded", completionTime=time.time())
            else:
                t.status.update(active=0, failed=1, phase="Failed", completionTime=time.time())
            self._emit("MODIFIED", t)
        # the devices stay reserved until the probe has seen them WITHOUT a new tenant
        if t.gpus and self._verify:
            self._verify_reclaim(t)
        with self._lock:
            if not t.deleted:
                self._release_gpus(t)
                self._schedule()
                self._respawn_hot(t)
        self._arm_ttl(t)

    # ------------------------------------------------------------ tasks that outlive the service
    # A task is its own session (its process group is what delete kills), so it survives a service that is killed --
    # and would keep its device while the restarted service, which knows nothing of it, hands the same device to the
    # next job.  Every running task therefore leaves a note under <log_dir>.running (pid + the kernel's start time of
    # that pid, so a recycled pid is never mistaken for it); a starting service kills what the notes of a DEAD service
    # still point to.  Notes of a service that is alive (two services sharing a log directory) are left alone.
    @staticmethod
    def _proc_start(pid):
        try:
            with open("/proc/%d/stat" % pid) as f:
                return f.read().rsplit(")", 1)[1].split()[19]       # field 22: starttime in clock ticks
        except (OSError, IndexError):
            return None

    def _register_running(self, t):
        try:
            d = self._run_dir
            os.makedirs(d, exist_ok=True)
            note = {"name": t.name, "pid": t.proc.pid, "start": self._proc_start(t.proc.pid), "gpus": list(t.gpus),
                    "owner": os.getpid(), "owner_start": self._proc_start(os.getpid())}
            tmp = os.path.join(d, t.name + ".json.tmp")
            with open(tmp, "w") as f:
                json.dump(note, f)
            os.replace(tmp, os.path.join(d, t.name + ".json"))
        except OSError:
            pass

    def _unregister_running(self, t):
        try:
            os.unlink(os.path.join(self._run_dir, t.name + ".json"))
        except OSError:
            pass

    def reap_orphans(self):
        """Kill the tasks a previous, dead service instance left running; returns their names."""
        d = self._run_dir
        killed = []
        for f in sorted(os.listdir(d)) if os.path.isdir(d) else []:
            if not f.endswith(".json"):
                continue
            path = os.path.join(d, f)
            try:
                with open(path) as fh:
                    note = json.load(fh)
                pid, owner = int(note["pid"]), int(note.get("owner", 0))
            except (OSError, ValueError, KeyError, TypeError):
                continue
            if owner and owner != os.getpid() and self._proc_start(owner) is not None \
                    and self._proc_start(owner) == note.get("owner_start"):
                log.warning("task %s belongs to a live service (pid %d) that shares %s: left alone", note.get("name"), owner,
                            self._log_dir)
                continue
            if owner == os.getpid() and self._proc_start(owner) == note.get("owner_start"):
                continue            # a second LocalCluster of this very process
            if note.get("start") is not None and self._proc_start(pid) == note["start"]:
                log.warning("task %s (pid %d, devices %s) outlived the service that started it: killing it",
                            note.get("name"), pid, note.get("gpus"))
                for sig, wait_s in ((signal.SIGTERM, self._grace), (signal.SIGKILL, self._kill_wait)):
                    try:
                        os.killpg(pid, sig)
                    except (ProcessLookupError, PermissionError):
                        try:
                            os.kill(pid, sig)
                        except (ProcessLookupError, PermissionError):
                            break
                    t_end = time.time() + wait_s
                    while time.time() < t_end and self._proc_start(pid) == note["start"]:
                        try:
                            os.waitpid(pid, os.WNOHANG)       # our own child in tests; a stranger's in production
                        except (ChildProcessError, OSError):
                            pass
                        time.sleep(0.05)
                    if self._proc_start(pid) != note["start"]:
                        break
                killed.append(note.get("name"))
            try:
                os.unlink(path)
            except OSError:
                pass
        return killed

    def _arm_ttl(self, t):
        ttl = (t.body.get("spec") or {}).get("ttlSecondsAfterFinished")
        try:
            ttl = float(self._finished_ttl if ttl is None else ttl)
        except (TypeError, ValueError):
            ttl = 0.0
        if ttl <= 0 or t.deleted or self._closed:
            return

        def expire():
            with self._lock:
                gone = self._closed or t.deleted or self._jobs.get((t.namespace, t.name)) is not t
            if gone:
                return
            try:
                self.delete_namespaced_job(t.name, t.namespace)
                log.info("finished job %s deleted after its ttl of %.0f s", t.name, ttl)
            except ApiException:
                pass

        timer = threading.Timer(ttl, expire)
        timer.daemon = True
        timer.name = "job-ttl-" + t.name
        timer.start()

    def _respawn_hot(self, t):
        # caller holds the lock: a released device that nobody took gets a fresh GPU-hot rank
        if self._hot is not None and not self._closed:
            for g in t.gpus:
                if g in self._free_gpus:
                    self._hot.spawn(g)

    @staticmethod
    def _sweep_stage(t):
        """A task removes its hdfs:// stage directory (``entry.py``) itself -- unless it was killed.  The launcher
        sweeps what a dead task left: ``<tmp>/dtf_hdfs_<task name>_*``."""
        import shutil
        import tempfile

        for d in glob.glob(os.path.join(tempfile.gettempdir(), "dtf_hdfs_%s_*" % t.name)):
            shutil.rmtree(d, ignore_errors=True)

    def _deadline_exceeded(self, t):
        """activeDeadlineSeconds: a task still running past its deadline is killed and reported
        failed (reason DeadlineExceeded) -- the heartbeat-free way to catch ranks hung in a collective."""
        if t.proc is None or t.proc.poll() is not None or t.deleted:
            return
        log.warning("task %s exceeded its deadline; killing", t.name)
        t.status["message"] = "DeadlineExceeded"
        self._kill(t)

    def _gang_task_gone(self, t):
        # caller holds the lock.  A task of a gang was deleted: the job is being torn down (cleanup, cancel, rollback
        # of a half-created job).  Devices set aside for tasks that never started go back, and tasks of this job that
        # arrive later are scheduled one by one.  The entry is dropped with the job's last task.
        gang = getattr(t, "gang", None)
        if not gang:
            return
        left = self._gangs.pop(gang, None)
        if left:
            self._free_gpus.extend(left)
            self._free_gpus.sort()
        if any(getattr(o, "gang", None) == gang for o in self._jobs.values()):
            self._gang_broken.add(gang)
        else:
            self._gang_broken.discard(gang)

    def _release_gpus(self, t):
        # caller holds the lock.  Idempotent per task: a second call must not re-add a device that has been
        # handed to another task in the meantime (double release -> two tenants on one GPU)
        if t.released:
            return
        t.released = True
        for g in t.gpus:
            assert g not in self._free_gpus, "GPU %d released twice" % g
            self._free_gpus.append(g)
        self._free_gpus.sort()

    def read_namespaced_job(self, name, namespace):
        with self._lock:
            t = self._jobs.get((namespace, name))
            if t is None:
                raise ApiException(404, "job %s not found" % name)
            return AttrDict.wrap(self._snapshot(t))

    def read_namespaced_pod_log(self, name, namespace=None, tail_lines=None, limit_bytes=4 << 20):
        """``kubectl logs``: the task's stdout + stderr (one file per task under ``launcher.log_dir``; it outlives the
        task, so the log of a reaped or failed task is still there).  ``tail_lines``: the last N lines only."""
        if os.sep in name or name.startswith("."):
            raise ApiException(400, "bad task name %r" % name)
        path = os.path.join(self._log_dir, name + ".log")
        if not os.path.isfile(path):
            raise ApiException(404, "no log for task %s" % name)
        size = os.path.getsize(path)
        with open(path, "rb") as f:
            if size > limit_bytes:
                f.seek(size - limit_bytes)
            text = f.read().decode(errors="replace")
        if tail_lines is not None:
            text = "".join(text.splitlines(True)[-max(0, int(tail_lines)):]) if int(tail_lines) > 0 else ""
        return text

    def list_namespaced_job(self, namespace):
        with self._lock:
            return AttrDict.wrap({
                "metadata": {"resourceVersion": str(self._rv)},
                "items": [self._snapshot(t) for (ns, _), t in self._jobs.items() if ns == namespace],
            })

    def _kill(self, t):
        """SIGTERM -> grace -> SIGKILL on the task's process group."""
        p = t.proc
        if p is None or p.poll() is not None:
            return "exited"
        how = "SIGTERM"
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            p.wait(self._grace)
        except subprocess.TimeoutExpired:
            how = "SIGKILL"
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            try:
                p.wait(self._kill_wait)
            except subprocess.TimeoutExpired:
                how = "unkillable"
        return how

    def _verify_reclaim(self, t):
        """After a GPU task ended (killed OR finished normally): is its HBM back?  Authoritative check = a fresh
        process asking the driver (``hipMemGetInfo`` through ``torch.cuda.mem_get_info``) on exactly the devices the
        task held, compared with what those devices showed when idle (the probe at service start, refreshed by every
        clean reclaim) + ``hbm_reclaim_slack_mb``.  Called while the devices are still reserved, so no new tenant
        can be mistaken for a leak.  The sysfs / rocm-smi counters are recorded only when they carry information
        (on some boxes they read 0 whatever is allocated)."""
        if not t.gpus:
            return None
        slack = self._cfg.getint("launcher", "hbm_reclaim_slack_mb", 64) << 20
        rep = {"task": t.name, "gpus": list(t.gpus), "exit_code": t.exit_code}
        if t.hbm_during:
            rep["during"] = t.hbm_during
        if t.hbm_before and any(v > 0 for v in t.hbm_before.values()):
            deadline = time.time() + 5.0
            while True:
                used = hbm_used_bytes()
                after = {g: used[g] for g in t.hbm_before if g < len(used)}
                ok = all(after[g] <= t.hbm_before[g] + slack for g in after)
                if ok or time.time() > deadline:
                    break
                time.sleep(0.1)
            rep.update(counter_before=t.hbm_before, counter_after=after, counter_ok=ok)
        probe = probe_hbm(t.gpus, self._python)
        if probe is not None:
            rep["probe"] = probe
            with self._lock:
                base = {p["gpu"]: self._idle_used.get(p["gpu"]) for p in probe}
            rep["idle_baseline"] = base
            checks = []
            for p in probe:
                b = base.get(p["gpu"])
                if b is None:
                    # no idle baseline (service started with the device busy): fall back to "far below what the
                    # task held" when a during-sample exists, else to the driver-reservation bound
                    held = next((d["used"] for d in (t.hbm_during or []) if d.get("gpu") == p["gpu"]), None)
                    checks.append(p["used"] <= (min(held // 4, 2 << 30) if held else (2 << 30)))
                else:
                    checks.append(p["used"] <= b + slack)
            rep["reclaimed"] = all(checks)
            if rep["reclaimed"]:
                with self._lock:
                    for p in probe:
                        self._idle_used[p["gpu"]] = min(p["used"], self._idle_used.get(p["gpu"], p["used"]))
        else:
            rep["reclaimed"] = rep.get("counter_ok")
        self.reclaim_reports.append(rep)
        log.info("HBM reclaim %s", rep)
        return rep

    def delete_namespaced_job(self, name, namespace, body=None):
        with self._lock:
            t = self._jobs.get((namespace, name))
            if t is None:
                raise ApiException(404, "job %s not found" % name)
            already = t.deleted
            t.deleted = True
            released = t.released
        if already:
            raise ApiException(409, "job %s is already being deleted" % name)
        how = self._kill(t)
        self._sweep_stage(t)
        # a task that ended on its own has been verified by its waiter thread already
        rep = self._verify_reclaim(t) if (self._verify and not released and t.gpus) else None
        with self._lock:
            self._jobs.pop((namespace, name), None)
            self._release_gpus(t)
            self._gang_task_gone(t)
            t.status.update(active=0, phase="Deleted")
            self._emit("DELETED", t)
            self._schedule()
            self._respawn_hot(t)
        return AttrDict.wrap({"status": "Success",
                              "details": {"name": name, "kind": "jobs", "signal": how, "hbm": rep}})

    # ---------------------------------------------------------------- watch
    def watch(self, namespace, resource_version=0, timeout=None, stop=None):
        """Generator of ``{"type","object"}`` events newer than
        ``resource_version`` (0 = replay everything, like a fresh k8s watch
        replays ADDED for existing objects).  ``timeout=None`` blocks forever
        (the reference's ``timeout_seconds=0``)."""
        rv = int(resource_version)
        deadline = None if timeout is None else time.monotonic() + timeout
        while True:
            with self._cv:
                batch = self._newer_than(rv)
                if not batch:
                    if self._closed or (stop is not None and stop.is_set()):
                        return
                    left = 0.25 if deadline is None else min(0.25, deadline - time.monotonic())
                    if left <= 0:
                        return
                    self._cv.wait(left)
                    batch = self._newer_than(rv)
            for erv, typ, snap in batch:
                rv = erv
                if snap["metadata"].get("namespace") != namespace:
                    continue
                yield {"type": typ, "object": AttrDict.wrap(copy.deepcopy(snap))}

    def _newer_than(self, rv):
        # caller holds the lock; events are sorted by resourceVersion (the log may have been compacted)
        import bisect

        i = bisect.bisect_right(self._events, rv, key=lambda e: e[0])
        return self._events[i:]

    def events_since(self, namespace, resource_version=0, wait=0.0):
        """Non-generator form used by the HTTP API (long poll)."""
        rv = int(resource_version)
        deadline = time.monotonic() + wait
        with self._cv:
            while True:
                out = [{"type": typ, "object": snap} for erv, typ, snap in self._events
                       if erv > rv and snap["metadata"].get("namespace") == namespace]
                left = deadline - time.monotonic()
                if out or left <= 0 or self._closed:
                    return out, self._rv
                self._cv.wait(min(left, 0.25))

    # ------------------------------------------------------------- shutdown
    def shutdown(self):
        with self._lock:
            self._closed = True
            tasks = list(self._jobs.values())
            self._cv.notify_all()
        if self._warm is not None:
            self._warm.close()
        if self._hot is not None:
            self._hot.close()
        for t in tasks:
            t.deleted = True
            self._kill(t)

    def free_gpus(self):
        with self._lock:
            return list(self._free_gpus)


# ---------------------------------------------------------------------------
# Process-wide default cluster + HTTP client with the same surface
# ---------------------------------------------------------------------------

_default = None
_default_lock = threading.Lock()


def install_cluster(cluster):
    global _default
    _default = cluster


def get_cluster():
    """The in-process :class:`LocalCluster` if one is installed (all-in-one
    service), else an HTTP :class:`ClusterClient` to ``[service] host:port``
    (the watcher / reaper daemons run as separate processes like the
    reference's ``tf_cleaner.py`` / ``remove_ps.py``)."""
    global _default
    with _default_lock:
        if _default is None:
            cfg = ApiConfig()
            _default = ClusterClient("http://%s:%d" % (cfg.get("service", "host", "127.0.0.1"),
                                                       cfg.getint("service", "port")))
        return _default


class ClusterClient:
    """k8s-shaped REST client for the node agent served by ``service/api.py``."""

    def __init__(self, base_url, timeout=30.0):
        self._base = base_url.rstrip("/")
        self._timeout = timeout

    def _req(self, method, path, body=None, timeout=None):
        import urllib.error
        import urllib.request

        data = None if body is None else json.dumps(body).encode()
        req = urllib.request.Request(self._base + path, data=data, method=method,
                                     headers={"Content-Type": "application/json"})
        try:
            with urllib.request.urlopen(req, timeout=timeout or self._timeout) as r:
                return json.loads(r.read().decode() or "null")
        except urllib.error.HTTPError as e:
            raise ApiException(e.code, e.read().decode(errors="replace"))
        except urllib.error.URLError as e:
            raise ApiException(503, str(e.reason))

    def create_namespaced_service(self, namespace, body):
        return AttrDict.wrap(self._req("POST", "/api/v1/namespaces/%s/services" % namespace, body))

    def delete_namespaced_service(self, name, namespace, body=None):
        return AttrDict.wrap(self._req("DELETE", "/api/v1/namespaces/%s/services/%s" % (namespace, name)))

    def create_namespaced_job(self, namespace, body):
        return AttrDict.wrap(self._req("POST", "/apis/batch/v1/namespaces/%s/jobs" % namespace, body))

    def delete_namespaced_job(self, name, namespace, body=None):
        return AttrDict.wrap(self._req("DELETE", "/apis/batch/v1/namespaces/%s/jobs/%s" % (namespace, name),
                                       timeout=120))

    def read_namespaced_job(self, name, namespace):
        return AttrDict.wrap(self._req("GET", "/apis/batch/v1/namespaces/%s/jobs/%s" % (namespace, name)))

    def read_namespaced_pod_log(self, name, namespace=None, tail_lines=None):
        q = "" if tail_lines is None else "?tailLines=%d" % int(tail_lines)
        return self._req("GET", "/api/v1/namespaces/%s/pods/%s/log%s" % (namespace or "default", name, q))["log"]

    def list_namespaced_job(self, namespace):
        return AttrDict.wrap(self._req("GET", "/apis/batch/v1/namespaces/%s/jobs" % namespace))

    def watch(self, namespace, resource_version=0, timeout=None, stop=None):
        """Long-poll watch with resourceVersion resume and reconnect (both are
        missing from the reference's watch loop)."""
        rv = int(resource_version)
        deadline = None if timeout is None else time.monotonic() + timeout
        backoff = 0.1
        while stop is None or not stop.is_set():
            if deadline is not None and time.monotonic() >= deadline:
                return
            try:
                r = self._req("GET", "/apis/batch/v1/namespaces/%s/jobs?watch=1&resourceVersion=%d&wait=1"
                              % (namespace, rv), timeout=15)
                backoff = 0.1
            except ApiException as e:
                log.warning("watch reconnect after error: %s", e)
                time.sleep(backoff)
                backoff = min(backoff * 2, 2.0)
                continue
            for ev in r["events"]:
                rv = max(rv, int(ev["object"]["metadata"]["resourceVersion"]))
                yield {"type": ev["type"], "object": AttrDict.wrap(ev["object"])}
