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
