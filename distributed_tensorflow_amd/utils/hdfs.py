"""WebHDFS client for the job spec's ``file`` / ``data`` / ``export`` fields (external X9, HDFS).

The reference hands ``[hdfs] web`` and ``[hdfs] namenode`` to ``/notebooks/entry.sh`` as argv[8], argv[9]
(``/root/reference/handlers/train_handler.py:80-83``, ``/root/reference/conf/config:19-26``) and the script inside its
image pulls the user's training file and data from HDFS through them.  Here the task entry point
(``service/entry.py``) does that itself: a field of the form ``hdfs://<namenode>/<path>`` (or ``webhdfs://``) is
fetched through the WebHDFS REST gateway before the task starts, ``export`` of that form is uploaded when the chief
finishes; plain paths stay local files.  Standard library only (``urllib``), so a task needs no extra package:

    GET  {web}/webhdfs/v1{path}?op=OPEN&user.name=U          (follows the 307 redirect to a datanode)
    GET  {web}/webhdfs/v1{path}?op=LISTSTATUS / GETFILESTATUS
    PUT  {web}/webhdfs/v1{path}?op=MKDIRS
    PUT  {web}/webhdfs/v1{path}?op=CREATE&overwrite=true     (two steps: redirect, then the data)
"""
from __future__ import annotations

import json
import os
import shutil
import urllib.error
import urllib.parse
import urllib.request

SCHEMES = ("hdfs", "webhdfs")


class HdfsError(RuntimeError):
    pass


def is_remote(path):
    return urllib.parse.urlparse(path or "").scheme in SCHEMES


def split(path):
    """``hdfs://nn:8020/a/b`` -> ``("nn:8020", "/a/b")``; a plain ``/a/b`` -> ``("", "/a/b")``."""
    u = urllib.parse.urlparse(path)
    if u.scheme in SCHEMES:
        return u.netloc, u.path or "/"
    return "", path


class _NoRedirect(urllib.request.HTTPRedirectHandler):
    def redirect_request(self, req, fp, code, msg, headers, newurl):      # CREATE answers 307 with the datanode URL
        return None


class WebHdfs:
    def __init__(self, web, user="root", timeout=10.0):
        """``web``: the gateway, ``http://host:14000`` or ``http://host:14000/webhdfs/v1`` (both spellings are in the
        reference's config: ``[hdfs] web`` and ``[request] hdfs_url``)."""
        web = (web or "").rstrip("/")
        if not web:
            raise HdfsError("no WebHDFS endpoint configured ([hdfs] web)")
        self.base = web if web.endswith("/webhdfs/v1") else web + "/webhdfs/v1"
        self.user, self.timeout = user, float(timeout)

    # -- plumbing -----------------------------------------------------------
    def _url(self, path, op, **params):
        q = {"op": op, "user.name": self.user}
        q.update(params)
        return self.base + urllib.parse.quote(path) + "?" + urllib.parse.urlencode(q)

    def _open(self, url, method="GET", data=None, redirect=True, length=None):
        req = urllib.request.Request(url, data=data, method=method)
        if data is not None:
            req.add_header("Content-Type", "application/octet-stream")
        if length is not None:
            req.add_header("Content-Length", str(int(length)))
        opener = urllib.request.build_opener() if redirect else urllib.request.build_opener(_NoRedirect)
        try:
            return opener.open(req, timeout=self.timeout)
        except urllib.error.HTTPError as e:
            if not redirect and e.code in (301, 302, 303, 307, 308):
                return e
            body = e.read(500).decode("utf-8", "replace")
            raise HdfsError("WebHDFS %s %s -> HTTP %d %s" % (method, url.split("?")[0], e.code, body)) from None
        except (urllib.error.URLError, OSError) as e:
            raise HdfsError("WebHDFS %s %s failed: %s" % (method, url.split("?")[0], e)) from None

    # -- operations ---------------------------------------------------------
    def status(self, path):
        """FileStatus dict (``type``: FILE | DIRECTORY, ``length``, ...) or ``None`` when the path does not exist."""
        try:
            with self._open(self._url(path, "GETFILESTATUS")) as r:
                return json.load(r)["FileStatus"]
        except HdfsError as e:
            if "HTTP 404" in str(e):
                return None
            raise

    def listdir(self, path):
        with self._open(self._url(path, "LISTSTATUS")) as r:
            return json.load(r)["FileStatuses"]["FileStatus"]

    def get_file(self, path, dest):
        os.makedirs(os.path.dirname(os.path.abspath(dest)) or ".", exist_ok=True)
        tmp = dest + ".part"
        with self._open(self._url(path, "OPEN")) as r, open(tmp, "wb") as f:
            shutil.copyfileobj(r, f, 1 << 20)
        os.replace(tmp, dest)
        return dest

    @staticmethod
    def _safe_name(name, what):
        """A directory entry reported by the gateway becomes a LOCAL file name: it must be one path component.
        ``..`` or a name with a separator would let a gateway write outside the task's stage directory."""
        if not name or name in (".", "..") or "/" in name or "\\" in name or "\0" in name:
            raise HdfsError("refusing %s %r from the WebHDFS gateway: not a plain file name" % (what, name))
        return name

    def get(self, path, dest_dir):
        """File -> ``dest_dir/<basename>``; directory -> ``dest_dir/<basename>/...`` (recursive).  Returns the local path."""
        st = self.status(path)
        if st is None:
            raise HdfsError("hdfs path %s does not exist" % path)
        base = os.path.basename(path.rstrip("/"))
        local = os.path.join(dest_dir, self._safe_name(base, "path") if base else "root")
        if st.get("type") == "DIRECTORY":
            os.makedirs(local, exist_ok=True)
            for ent in self.listdir(path):
                self.get(path.rstrip("/") + "/" + self._safe_name(ent.get("pathSuffix", ""), "pathSuffix"), local)
            return local
        return self.get_file(path, local)

    def mkdirs(self, path):
        with self._open(self._url(path, "MKDIRS"), "PUT") as r:
            return bool(json.load(r).get("boolean"))

    def put_file(self, local, path):
        first = self._open(self._url(path, "CREATE", overwrite="true"), "PUT", redirect=False)
        loc = first.headers.get("Location") if first is not None else None
        if hasattr(first, "close"):
            first.close()
        if not loc:          # a gateway that takes the data in one step (HttpFS with data=true) answers 201 directly
            loc = self._url(path, "CREATE", overwrite="true", data="true")
        # the body is streamed from the file (checkpoint shards are multi-GB): urllib sends a file object in blocks
        # when Content-Length is given
        with open(local, "rb") as f:
            with self._open(loc, "PUT", data=f, length=os.path.getsize(local)) as r:
                if r.status not in (200, 201):
                    raise HdfsError("WebHDFS CREATE %s -> HTTP %d" % (path, r.status))
        return path

    def put(self, local, path):
        """Upload a file, or a directory tree, to ``path``."""
        if os.path.isdir(local):
            self.mkdirs(path)
            for name in sorted(os.listdir(local)):
                self.put(os.path.join(local, name), path.rstrip("/") + "/" + name)
            return path
        return self.put_file(local, path)


class ExportSync:
    """A task's local ``export`` directory mirrored to ``hdfs://`` (``entry.py``).

    * :meth:`fetch_latest` (before training, when the job resumes): the remote ``checkpoint`` pointer and every file
      of the step it names -- without this a resumed job would look at an empty stage directory and start over;
    * :meth:`push` after EVERY checkpoint and once more when the task ends, whatever its exit code (mid-run
      checkpoints exist for the runs that die): data files first, then -- ``publish=True``, the chief after a
      barrier -- the index and the pointer, so a reader that follows the pointer always finds the shards."""

    POINTER = "checkpoint"

    def __init__(self, client, local_dir, remote_dir):
        self.client, self.local, self.remote = client, local_dir, remote_dir.rstrip("/") or "/"
        self._sent = {}
        self._made = False

    def fetch_latest(self):
        """-> step fetched, or None when the remote export holds no checkpoint."""
        ptr = self.remote + "/" + self.POINTER
        if self.client.status(ptr) is None:
            return None
        os.makedirs(self.local, exist_ok=True)
        self.client.get_file(ptr, os.path.join(self.local, self.POINTER))
        prefix = None
        with open(os.path.join(self.local, self.POINTER)) as f:
            for line in f:
                if line.startswith("model_checkpoint_path") and '"' in line:
                    prefix = line.split('"')[1]
        if not prefix:
            return None
        prefix = WebHdfs._safe_name(prefix, "checkpoint prefix")
        for ent in self.client.listdir(self.remote):
            name = WebHdfs._safe_name(ent.get("pathSuffix", ""), "pathSuffix")
            if ent.get("type") == "FILE" and name.startswith(prefix + "."):
                dst = os.path.join(self.local, name)
                self.client.get_file(self.remote + "/" + name, dst)
                st = os.stat(dst)
                self._sent[name] = (st.st_mtime_ns, st.st_size)       # already remote: do not send it back
        st = os.stat(os.path.join(self.local, self.POINTER))
        self._sent[self.POINTER] = (st.st_mtime_ns, st.st_size)
        return int(prefix.rsplit("-", 1)[1]) if "-" in prefix else None

    def _is_publish_file(self, name):
        return name == self.POINTER or name.endswith(".index")

    def push(self, publish=False):
        """Upload what changed since the last call.  ``publish=False``: everything but index + pointer."""
        if not os.path.isdir(self.local):
            return []
        names = sorted(n for n in os.listdir(self.local) if os.path.isfile(os.path.join(self.local, n)))
        # data, replicated state, then the index, the pointer last
        order = [n for n in names if not self._is_publish_file(n)]
        if publish:
            order += [n for n in names if n.endswith(".index")] + [n for n in names if n == self.POINTER]
        done = []
        for n in order:
            p = os.path.join(self.local, n)
            st = os.stat(p)
            sig = (st.st_mtime_ns, st.st_size)
            if self._sent.get(n) == sig:
                continue
            if not self._made:
                self.client.mkdirs(self.remote)
                self._made = True
            self.client.put_file(p, self.remote + "/" + n)
            self._sent[n] = sig
            done.append(n)
        return done


def client_from(web=None, user=None, cfg=None):
    """Gateway from the entry point's argv[8] (``[hdfs] web``), falling back to ``[request] hdfs_url``."""
    if cfg is None:
        from .config import ApiConfig

        cfg = ApiConfig()
    if not web:
        web = cfg.get("hdfs", "web") if cfg.has_option("hdfs", "web") else ""
    if not web and cfg.has_option("request", "hdfs_url"):
        web = cfg.get("request", "hdfs_url")
    if user is None:
        user = cfg.get("request", "hdfs_user") if cfg.has_option("request", "hdfs_user") else "root"
    timeout = cfg.getfloat("request", "timeout") if cfg.has_option("request", "timeout") else 10.0
    return WebHdfs(web, user, timeout)


def localize(path, dest_dir, web=None, cfg=None):
    """``hdfs://`` path -> fetched local copy under ``dest_dir``; anything else is returned unchanged."""
    if not is_remote(path):
        return path
    _, p = split(path)
    return client_from(web, cfg=cfg).get(p, dest_dir)
