"""INI configuration singleton (component C7, reference ``util/ApiConfiger.py:1-26``).

Same surface as the reference -- ``ApiConfig().get/getint/getfloat/getboolean
(section, key)`` -- with three differences that the reference's latent defects
call for:

* the file is parsed once and re-parsed only when its mtime changes (the
  reference re-reads ``conf/config`` on *every* ``ApiConfig()`` call);
* every key can be overridden from the environment as ``DTF_<SECTION>_<KEY>``
  so tests can run services on ephemeral ports in parallel;
* ``DTF_CONFIG`` selects another file, and ``ApiConfig.override`` patches values
  in-process (used by tests and by the all-in-one service).
"""
from __future__ import annotations

import configparser
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_CONFIG_PATH = os.path.normpath(os.path.join(_HERE, "..", "conf", "config"))
DEFAULT_LOGGING_CONF = os.path.normpath(os.path.join(_HERE, "..", "conf", "logging.conf"))


class ApiConfig:
    _instance = None
    _lock = threading.Lock()

    def __new__(cls, *args, **kwargs):
        with cls._lock:
            if cls._instance is None:
                inst = super().__new__(cls)
                inst._parser = None
                inst._mtime = None
                inst._path = None
                inst._overrides = {}
                cls._instance = inst
        return cls._instance

    def __init__(self):
        path = os.environ.get("DTF_CONFIG", DEFAULT_CONFIG_PATH)
        try:
            mtime = os.stat(path).st_mtime_ns
        except OSError:
            mtime = None
        if self._parser is None or path != self._path or mtime != self._mtime:
            parser = configparser.ConfigParser()
            if mtime is not None:
                parser.read(path)
            self._parser, self._path, self._mtime = parser, path, mtime

    # -- lookup -----------------------------------------------------------
    def _raw(self, section, key):
        env = os.environ.get("DTF_%s_%s" % (section.upper(), key.upper()))
        if env is not None:
            return env
        ov = self._overrides.get((section, key))
        if ov is not None:
            return ov
        return self._parser.get(section, key)

    def get(self, section, key, fallback=None):
        try:
            return self._raw(section, key)
        except (configparser.NoSectionError, configparser.NoOptionError):
            if fallback is not None:
                return fallback
            raise

    def getint(self, section, key, fallback=None):
        v = self.get(section, key, None if fallback is None else str(fallback))
        return int(v)

    def getfloat(self, section, key, fallback=None):
        v = self.get(section, key, None if fallback is None else str(fallback))
        return float(v)

    def getboolean(self, section, key, fallback=None):
        v = self.get(section, key, None if fallback is None else str(fallback))
        return str(v).strip().lower() in ("1", "true", "yes", "on")

    def has_option(self, section, key):
        try:
            self._raw(section, key)
            return True
        except (configparser.NoSectionError, configparser.NoOptionError):
            return False

    def sections(self):
        return self._parser.sections()

    # -- test / embedding helpers ----------------------------------------
    @classmethod
    def override(cls, section, key, value):
        cls()._overrides[(section, key)] = None if value is None else str(value)

    @classmethod
    def clear_overrides(cls):
        cls()._overrides.clear()


def setup_logging(log_dir=None):
    """``logging.config.fileConfig`` with the reference's format
    (``conf/logging.conf:10-21``); creates ``logs/`` like ``service/disTf.py:25-27``."""
    import logging
    import logging.config

    log_dir = log_dir or os.environ.get("DTF_LOG_DIR", "logs")
    os.makedirs(log_dir, exist_ok=True)
    cwd = os.getcwd()
    try:
        # the conf names a path relative to the log dir's parent
        os.chdir(os.path.dirname(os.path.abspath(log_dir)) or ".")
        if os.path.basename(os.path.abspath(log_dir)) != "logs":
            os.makedirs("logs", exist_ok=True)
        logging.config.fileConfig(DEFAULT_LOGGING_CONF, disable_existing_loggers=False)
    finally:
        os.chdir(cwd)
    return logging.getLogger()
