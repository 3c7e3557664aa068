"""WebHDFS access for the job spec's file / data / export fields (utils/hdfs.py, service/entry.py) against an
in-process fake gateway that behaves like HttpFS / a namenode: OPEN and CREATE answer with a 307 redirect to a
"datanode" URL, LISTSTATUS / GETFILESTATUS / MKDIRS answer JSON.  Parity unpinned: the reference holds no HDFS fixture
(its entry.sh lives inside an image that is not in the tree); the wire format is the public WebHDFS REST API."""
import json
import os
import subprocess
import sys
import threading
import urllib.parse
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer

import pytest

from distributed_tensorflow_amd.utils import hdfs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeHdfs(BaseHTTPRequestHandler):
    files = {}           # path -> bytes
    dirs = set()
    log = []

    def log_message(self, *a):
        pass

    def _parse(self):
        u = urllib.parse.urlparse(self.path)
        q = dict(urllib.parse.parse_qsl(u.query))
        assert u.path.startswith("/webhdfs/v1") or u.path.startswith("/dn/webhdfs/v1")
        datanode = u.path.startswith("/dn/")
        path = urllib.parse.unquote(u.path.split("/webhdfs/v1", 1)[1]) or "/"
        FakeHdfs.log.append((self.command, q.get("op"), path, q.get("user.name"), datanode))
        return q, path, datanode

    def _json(self, code, obj):
        body = json.dumps(obj).encode()
        self.send_response(code)
        self.send_header("Content-Type", "application/json")
        self.send_header("Content-Length", str(len(body)))
        self.end_headers()
        self.wfile.write(body)

    def _redirect(self):
        self.send_response(307)
        self.send_header("Location", "http://127.0.0.1:%d/dn%s" % (self.server.server_address[1], self.path))
        self.send_header("Content-Length", "0")
        self.end_headers()

    def _status(self, path):
        if path in self.files:
            return {"type": "FILE", "length": len(self.files[path]), "pathSuffix": os.path.basename(path)}
        if path.rstrip("/") in self.dirs or path == "/":
            return {"type": "DIRECTORY", "length": 0, "pathSuffix": os.path.basename(path.rstrip("/"))}
        return None

    def do_GET(self):
        q, path, datanode = self._parse()
        op = q.get("op")
        if op == "GETFILESTATUS":
            st = self._status(path)
            if st is None:
                return self._json(404, {"RemoteException": {"exception": "FileNotFoundException"}})
            return self._json(200, {"FileStatus": st})
        if op == "LISTSTATUS":
            base = path.rstrip("/") + "/"
            names = sorted({p[len(base):].split("/")[0] for p in list(self.files) + list(self.dirs) if p.startswith(base)})
            return self._json(200, {"FileStatuses": {"FileStatus": [self._status(base + n) for n in names]}})
        if op == "OPEN":
            if path not in self.files:
                return self._json(404, {"RemoteException": {"exception": "FileNotFoundException"}})
            if not datanode:
                return self._redirect()
            body = self.files[path]
            self.send_response(200)
            self.send_header("Content-Length", str(len(body)))
            self.end_headers()
            self.wfile.write(body)
            return
        self._json(400, {"RemoteException": {"exception": "UnsupportedOperationException"}})

    def do_PUT(self):
        q, path, datanode = self._parse()
        op = q.get("op")
        n = int(self.headers.get("Content-Length") or 0)
        data = self.rfile.read(n) if n else b""
        if op == "MKDIRS":
            FakeHdfs.dirs.add(path.rstrip("/"))
            return self._json(200, {"boolean": True})
        if op == "CREATE":
            if not datanode:
                return self._redirect()
            FakeHdfs.files[path] = data
            self.send_response(201)
            self.send_header("Content-Length", "0")
            self.end_headers()
            return
        self._json(400, {"RemoteException": {"exception": "UnsupportedOperationException"}})


@pytest.fixture()
def gateway():
    FakeHdfs.files = {"/jobs/train.py": b"print('hello')\n", "/data/set1/a.bin": b"A" * 3000,
                      "/data/set1/sub/b.txt": b"bee"}
    FakeHdfs.dirs = {"/jobs", "/data", "/data/set1", "/data/set1/sub"}
    FakeHdfs.log = []
    srv = ThreadingHTTPServer(("127.0.0.1", 0), FakeHdfs)
    t = threading.Thread(target=srv.serve_forever, daemon=True)
    t.start()
    yield "http://127.0.0.1:%d" % srv.server_address[1]
    srv.shutdown()
    srv.server_close()


def test_paths_and_schemes():
    assert hdfs.is_remote("hdfs://nn:8020/a/b") and hdfs.is_remote("webhdfs://nn/a")
    assert not hdfs.is_remote("/local/path") and not hdfs.is_remote("builtin://resnet50?steps=3") and not hdfs.is_remote("")
    assert hdfs.split("hdfs://nn:8020/a/b") == ("nn:8020", "/a/b")
    assert hdfs.split("/a/b") == ("", "/a/b")


def test_get_file_directory_and_put_round_trip(gateway, tmp_path):
    c = hdfs.WebHdfs(gateway, user="alice", timeout=5)
    assert c.status("/jobs/train.py")["type"] == "FILE" and c.status("/nope") is None
    f = c.get("/jobs/train.py", str(tmp_path / "f"))
    assert open(f, "rb").read() == b"print('hello')\n" and f.endswith("train.py")
    d = c.get("/data/set1", str(tmp_path / "d"))
    assert open(os.path.join(d, "a.bin"), "rb").read() == b"A" * 3000
    assert open(os.path.join(d, "sub", "b.txt"), "rb").read() == b"bee"
    # the gateway spelling of [request] hdfs_url works too, the user name travels, OPEN went through the datanode hop
    c2 = hdfs.WebHdfs(gateway + "/webhdfs/v1", user="alice")
    assert c2.status("/data")["type"] == "DIRECTORY"
    assert all(e[3] == "alice" for e in FakeHdfs.log)
    assert ("GET", "OPEN", "/jobs/train.py", "alice", True) in FakeHdfs.log
    out = tmp_path / "export"
    (out / "ckpt").mkdir(parents=True)
    (out / "ckpt" / "model.index").write_bytes(b"idx")
    (out / "metrics.json").write_text("{}")
    c.put(str(out), "/out/run1")
    assert FakeHdfs.files["/out/run1/ckpt/model.index"] == b"idx" and FakeHdfs.files["/out/run1/metrics.json"] == b"{}"
    assert "/out/run1/ckpt" in FakeHdfs.dirs
    with pytest.raises(hdfs.HdfsError):
        c.get("/missing", str(tmp_path / "x"))
    with pytest.raises(hdfs.HdfsError):
        hdfs.WebHdfs("http://127.0.0.1:9", timeout=1).status("/a")          # nothing listens there


def test_task_entry_fetches_file_and_data_and_uploads_export(gateway, tmp_path):
    """The argv contract of the reference's entry.sh with hdfs:// fields: the script and the data set arrive through
    the gateway of argv[8], the export directory is uploaded when the script has run."""
    FakeHdfs.files["/jobs/user_train.py"] = (
        b"import os, sys\n"
        b"data, export = sys.argv[6], sys.argv[7]\n"
        b"n = len(open(os.path.join(data, 'a.bin'), 'rb').read())\n"
        b"open(os.path.join(export, 'result.txt'), 'w').write('%s %s %d' % (os.environ['DTF_JOB_NAME'], os.environ['RANK'], n))\n")
    env = dict(os.environ, PYTHONPATH=ROOT, DTF_TASK_NAME="tf-abc-worker-0-1")
    r = subprocess.run([sys.executable, "-m", "distributed_tensorflow_amd.service.entry", "worker", "0", "", "127.0.0.1:29517",
                        "hdfs://nn/jobs/user_train.py", "hdfs://nn/data/set1", "hdfs://nn/out/run7", gateway, "hdfs://nn"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert FakeHdfs.files["/out/run7/result.txt"] == b"worker 0 3000"
    # a file that is not there fails the task cleanly
    r = subprocess.run([sys.executable, "-m", "distributed_tensorflow_amd.service.entry", "worker", "0", "", "127.0.0.1:29517",
                        "hdfs://nn/jobs/absent.py", "/tmp", "/tmp", gateway, "hdfs://nn"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "does not exist" in r.stderr


def test_gateway_supplied_names_cannot_escape_the_stage_directory(gateway, tmp_path):
    """``pathSuffix`` comes from the gateway and becomes a local file name.  ``..`` / names with a
    separator are refused instead of being joined into a path outside the destination."""
    c = hdfs.WebHdfs(gateway)
    for bad in ("..", "../evil", "a/b", ""):
        with pytest.raises(hdfs.HdfsError):
            hdfs.WebHdfs._safe_name(bad, "pathSuffix")

    class Evil(hdfs.WebHdfs):
        def listdir(self, path):
            return [{"pathSuffix": "../../escaped.txt", "type": "FILE"}]

    e = Evil(gateway)
    with pytest.raises(hdfs.HdfsError):
        e.get("/data/set1", str(tmp_path / "stage"))
    assert not (tmp_path / "escaped.txt").exists()
    assert c.get("/data/set1", str(tmp_path / "ok"))


def test_put_streams_the_file_with_a_content_length(gateway, tmp_path):
    big = tmp_path / "shard.bin"
    big.write_bytes(os.urandom(3 << 20))
    c = hdfs.WebHdfs(gateway)
    c.mkdirs("/ck")
    c.put_file(str(big), "/ck/shard.bin")
    assert FakeHdfs.files["/ck/shard.bin"] == big.read_bytes()


def _entry(gateway, argv_file, export, task="tf-3e8f3702-d10b-11e8-abe4-fa163ef8da8a-worker-0-1", extra_env=None):
    env = dict(os.environ, PYTHONPATH=ROOT, DTF_TASK_NAME=task, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-m", "distributed_tensorflow_amd.service.entry", "worker", "0", "",
                           "127.0.0.1:29518", argv_file, "/tmp", export, gateway, "hdfs://nn"],
                          env=env, capture_output=True, text=True, timeout=300)


def test_remote_export_checkpoints_during_the_run_resume_and_failure(gateway, tmp_path):
    """(a) every ``ckpt_every`` checkpoint reaches hdfs:// while the job runs -- data files before the
    index and the pointer; (b) a task that FAILS still leaves its earlier checkpoints remote; (c) ``resume=1`` with a
    remote export first fetches the latest checkpoint; (d) the stage directory is gone when the task ends."""
    import glob
    import tempfile

    stage_glob = os.path.join(tempfile.gettempdir(), "dtf_hdfs_tf-3e8f3702*")
    # a run that checkpoints at 4 and 8 and dies (injected exception) at step 10
    r = _entry(gateway, "builtin://mnist_mlp?steps=12&batch=16&ckpt_every=4&log_every=4&fault=raise:worker=0,step=10",
               "hdfs://nn/exp/runA")
    assert r.returncode == 1 and "injected fault" in r.stderr, r.stdout + r.stderr
    names = sorted(p[len("/exp/runA/"):] for p in FakeHdfs.files if p.startswith("/exp/runA/"))
    assert "checkpoint" in names and "model.ckpt-8.index" in names and "model.ckpt-4.index" in names, names
    assert any(n.startswith("model.ckpt-8.data-") for n in names), names
    assert b"model.ckpt-8" in FakeHdfs.files["/exp/runA/checkpoint"]
    puts = [e[2] for e in FakeHdfs.log if e[0] == "PUT" and e[1] == "CREATE" and e[4]]
    i_data = max(i for i, p in enumerate(puts) if p.endswith("model.ckpt-8.data-00000-of-00001"))
    i_index = max(i for i, p in enumerate(puts) if p.endswith("model.ckpt-8.index"))
    i_ptr = max(i for i, p in enumerate(puts) if p.endswith("/checkpoint"))
    assert i_data < i_index < i_ptr, puts
    assert glob.glob(stage_glob) == [], "stage directory leaked"
    # the same export, resumed: starts from step 8, finishes at 12
    r = _entry(gateway, "builtin://mnist_mlp?steps=12&batch=16&ckpt_every=4&log_every=4&resume=1", "hdfs://nn/exp/runA")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fetched checkpoint step 8" in r.stdout and "resumed from step 8" in r.stdout, r.stdout
    assert b"model.ckpt-12" in FakeHdfs.files["/exp/runA/checkpoint"]
    assert glob.glob(stage_glob) == []
    # resume against an empty remote export starts from scratch, cleanly
    r = _entry(gateway, "builtin://mnist_mlp?steps=4&batch=16&resume=1", "hdfs://nn/exp/fresh")
    assert r.returncode == 0 and "no checkpoint under hdfs:/exp/fresh" in r.stdout, r.stdout + r.stderr
