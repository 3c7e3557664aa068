"""Service entrypoint (component C1, reference ``service/disTf.py:1-33``).

``DisDeepService.start()`` serves ``/v1/train`` on ``[service] port`` (ref
``:17-22``).  Besides the reference's route it serves the node agent's
k8s-shaped REST API (``/api/v1/...services``, ``/apis/batch/v1/...jobs`` with a
long-poll watch) so that ``tf_cleaner`` and ``remove_ps`` can run as separate
daemons exactly like the reference's three-process deployment, and
``--all-in-one`` runs store + REST + watcher + reaper in one process (shortest
submit -> step-1 path: no extra hops, no second interpreter start).

    python -m distributed_tensorflow_amd.service.disTf --all-in-one
    curl -d @post-data http://127.0.0.1:12345/v1/train
"""
from __future__ import annotations

import asyncio
import json
import logging
import threading
import time

from aiohttp import web

from ..utils.config import ApiConfig, setup_logging
from ..utils.store import MemoryStore, RedisHelper, StoreServer
from . import manifest
from .launcher import ApiException, LocalCluster, install_cluster
from .remove_ps import PsKiller
from .tf_cleaner import Cleaner
from .train_handler import TrainHandler

log = logging.getLogger(__name__)


def _json(data, status=200):
    return web.Response(text=json.dumps(data), status=status, content_type="application/json")


class DisDeepService:
    def __init__(self, cluster=None, redis=None, cfg=None):
        self.config = cfg or ApiConfig()
        self.cluster = cluster
        self.redis = redis
        self.handler = TrainHandler(cluster=cluster, redis=redis, cfg=self.config)
        self._runner = None
        self._loop = None
        self._thread = None
        self.port = None

    # -- /v1/train --------------------------------------------------------------
    async def get_train(self, request):
        uid = request.query.get("uid")
        loop = asyncio.get_running_loop()
        res = await loop.run_in_executor(None, self.handler.status, uid)
        if res is None:
            return _json({"error": "unknown uid"}, 404)
        return _json(res)

    async def post_train(self, request):
        t_submit = time.time()
        body = await request.read()
        log.info("POST data: %s", body.decode(errors="replace"))
        loop = asyncio.get_running_loop()
        try:
            info = self.handler.parse(body)
            log.info("parse data: %s", info)
            names = await loop.run_in_executor(None, self.handler.submit, info, t_submit)
        except manifest.SpecError as e:
            return _json({"error": str(e)}, 400)
        except ApiException as e:
            return _json({"error": str(e)}, e.status if 400 <= e.status < 600 else 500)
        return _json(names)

    async def delete_train(self, request):
        uid = request.query.get("uid")
        if not uid:
            return _json({"error": "uid required"}, 400)
        loop = asyncio.get_running_loop()
        ok = await loop.run_in_executor(None, self.handler.cancel, uid)
        return _json({"cancelled": ok}, 200 if ok else 404)

    async def get_logs(self, request):
        """``GET /v1/train/logs?uid=<uid>[&task=worker-0][&tail=200]`` -> ``{task name: text}``: what ``kubectl logs``
        shows for the reference's pods.  The record names the job's tasks; the node agent keeps one file per task."""
        uid, task, tail = request.query.get("uid"), request.query.get("task"), request.query.get("tail")
        if not uid:
            return _json({"error": "uid required"}, 400)
        loop = asyncio.get_running_loop()
        try:
            res = await loop.run_in_executor(None, self.handler.logs, uid, task, int(tail) if tail else None)
        except ValueError:
            return _json({"error": "tail must be a number"}, 400)
        if res is None:
            return _json({"error": "unknown uid"}, 404)
        return _json(res)

    async def pod_log(self, request):
        tail = request.query.get("tailLines")
        loop = asyncio.get_running_loop()
        try:
            text = await loop.run_in_executor(None, self.cluster.read_namespaced_pod_log, request.match_info["name"],
                                              request.match_info["ns"], int(tail) if tail else None)
        except ApiException as e:
            return _json({"error": e.reason}, e.status)
        return _json({"log": text})

    # -- node agent API (k8s-shaped) -------------------------------------------
    async def _call(self, fn, *args):
        loop = asyncio.get_running_loop()
        try:
            return _json(await loop.run_in_executor(None, fn, *args))
        except ApiException as e:
            return _json({"error": e.reason}, e.status)

    async def svc_create(self, request):
        body = await request.json()
        return await self._call(self.cluster.create_namespaced_service, request.match_info["ns"], body)

    async def svc_delete(self, request):
        return await self._call(self.cluster.delete_namespaced_service,
                                request.match_info["name"], request.match_info["ns"])

    async def job_create(self, request):
        body = await request.json()
        return await self._call(self.cluster.create_namespaced_job, request.match_info["ns"], body)

    async def job_delete(self, request):
        return await self._call(self.cluster.delete_namespaced_job,
                                request.match_info["name"], request.match_info["ns"])

    async def job_read(self, request):
        return await self._call(self.cluster.read_namespaced_job,
                                request.match_info["name"], request.match_info["ns"])

    async def job_list(self, request):
        ns = request.match_info["ns"]
        if request.query.get("watch"):
            rv = int(request.query.get("resourceVersion", 0))
            wait = float(request.query.get("wait", 1))
            loop = asyncio.get_running_loop()
            events, cur = await loop.run_in_executor(None, self.cluster.events_since, ns, rv, wait)
            return _json({"events": events, "resourceVersion": cur})
        return await self._call(self.cluster.list_namespaced_job, ns)

    async def healthz(self, request):
        return _json({"ok": True, "free_gpus": self.cluster.free_gpus() if self.cluster else None})

    def make_app(self):
        app = web.Application()
        app.router.add_get("/v1/train", self.get_train)
        app.router.add_post("/v1/train", self.post_train)
        app.router.add_delete("/v1/train", self.delete_train)
        app.router.add_get("/v1/train/logs", self.get_logs)
        app.router.add_get("/healthz", self.healthz)
        if self.cluster is not None and isinstance(self.cluster, LocalCluster):
            app.router.add_post("/api/v1/namespaces/{ns}/services", self.svc_create)
            app.router.add_delete("/api/v1/namespaces/{ns}/services/{name}", self.svc_delete)
            app.router.add_post("/apis/batch/v1/namespaces/{ns}/jobs", self.job_create)
            app.router.add_get("/apis/batch/v1/namespaces/{ns}/jobs", self.job_list)
            app.router.add_get("/apis/batch/v1/namespaces/{ns}/jobs/{name}", self.job_read)
            app.router.add_delete("/apis/batch/v1/namespaces/{ns}/jobs/{name}", self.job_delete)
            app.router.add_get("/api/v1/namespaces/{ns}/pods/{name}/log", self.pod_log)
        return app

    # -- lifecycle ---------------------------------------------------------------
    async def _serve(self, host, port):
        self._runner = web.AppRunner(self.make_app(), access_log=None)
        await self._runner.setup()
        site = web.TCPSite(self._runner, host, port)
        await site.start()
        self.port = site._server.sockets[0].getsockname()[1]

    def start(self):
        """Blocking, like the reference's ``IOLoop.current().start()``."""
        host = self.config.get("service", "host", "127.0.0.1")
        port = self.config.getint("service", "port")
        self.reconcile()
        loop = asyncio.new_event_loop()
        asyncio.set_event_loop(loop)
        loop.run_until_complete(self._serve(host, port))
        log.info("service listening on %s:%d", host, self.port)
        print("dtf service listening on %s:%d" % (host, self.port), flush=True)
        try:
            loop.run_forever()
        except KeyboardInterrupt:
            pass
        finally:
            loop.run_until_complete(self._runner.cleanup())

    def reconcile(self):
        """Before serving: records left "running" by a previous instance of this service (external store)."""
        if isinstance(self.cluster, LocalCluster):
            try:
                return self.handler.reconcile()
            except Exception:       # noqa: BLE001 -- a store that is not up yet must not keep the service down
                log.exception("reconcile failed")
        return []

    def start_background(self, host=None, port=None):
        """Serve from a daemon thread; returns the bound port (tests, bench)."""
        self.reconcile()
        host = host or self.config.get("service", "host", "127.0.0.1")
        port = self.config.getint("service", "port") if port is None else port
        ready = threading.Event()

        def run():
            self._loop = asyncio.new_event_loop()
            asyncio.set_event_loop(self._loop)
            self._loop.run_until_complete(self._serve(host, port))
            ready.set()
            self._loop.run_forever()
            self._loop.run_until_complete(self._runner.cleanup())

        self._thread = threading.Thread(target=run, name="dtf-rest", daemon=True)
        self._thread.start()
        if not ready.wait(10):
            raise RuntimeError("REST service did not start")
        return self.port

    def stop(self):
        if self._loop is not None:
            self._loop.call_soon_threadsafe(self._loop.stop)
            self._thread.join(5)
            self._loop = None


class AllInOne:
    """store + node agent + REST + watcher + reaper in one process."""

    def __init__(self, gpus=None, log_dir=None, extra_env=None, store_port=None, serve_store=True):
        self.store = MemoryStore()
        RedisHelper.install(self.store)
        self.store_server = None
        env = dict(extra_env or {})
        if serve_store:
            # ranks report metrics through the RESP server
            self.store_server = StoreServer("127.0.0.1", store_port or 0, self.store).start()
            env.setdefault("DTF_REDIS_HOST", "127.0.0.1")
            env.setdefault("DTF_REDIS_PORT", str(self.store_server.port))
        self.cluster = LocalCluster(gpus=gpus, log_dir=log_dir, extra_env=env)
        install_cluster(self.cluster)
        self.service = DisDeepService(cluster=self.cluster, redis=self.store)
        self.cleaner = Cleaner(cluster=self.cluster)
        self.killer = PsKiller(cluster=self.cluster, redis=self.store)

    def start(self, port=None):
        self.cleaner.loadHandlers()
        for h in self.cleaner.eventHandlers:
            if hasattr(h, "_rc"):
                h._rc = self.store
        self.cleaner.start()
        self.killer.start()
        return self.service.start_background(port=port)

    def stop(self):
        self.killer.stop()
        self.cleaner.stop()
        self.service.stop()
        self.cluster.shutdown()
        if self.store_server:
            self.store_server.stop()
        RedisHelper.reset()
        install_cluster(None)


def main(argv=None):
    import argparse
    import os

    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--all-in-one", action="store_true",
                    help="also run the store, the watcher and the reaper in this process")
    ap.add_argument("--gpus", default=None, help="comma-separated device indices to schedule onto")
    a = ap.parse_args(argv)
    os.makedirs("logs", exist_ok=True)
    setup_logging()
    gpus = [int(x) for x in a.gpus.split(",")] if a.gpus else None
    if a.all_in_one:
        aio = AllInOne(gpus=gpus, store_port=ApiConfig().getint("redis", "port"))
        aio.cleaner.loadHandlers()
        aio.cleaner.start()
        aio.killer.start()
        try:
            aio.service.start()
        finally:
            aio.stop()
    else:
        cluster = LocalCluster(gpus=gpus)
        install_cluster(cluster)
        try:
            DisDeepService(cluster=cluster).start()
        finally:
            cluster.shutdown()


if __name__ == "__main__":
    main()
