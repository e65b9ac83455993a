"""absl-style logging shim (`logging.info`, `set_verbosity`) over the stdlib logger.

Log lines go to stderr, one stream per rank, as in the reference where each
process's stderr is redirected to `log$i.log` (`ps_server/run.sh:5`).
"""
from __future__ import annotations

import logging as _logging
import os
import sys

DEBUG, INFO, WARNING, ERROR, FATAL = (_logging.DEBUG, _logging.INFO, _logging.WARNING,
                                      _logging.ERROR, _logging.CRITICAL)

_logger = _logging.getLogger("dtf_amd")
_logger.propagate = False


class _Formatter(_logging.Formatter):
  def format(self, record):
    lvl = record.levelname[0]
    t = self.formatTime(record, "%m%d %H:%M:%S")
    rank = os.environ.get("RANK")
    tag = (" r%s" % rank) if rank is not None else ""
    return "%s%s.%06d %d%s %s:%d] %s" % (
        lvl, t, int((record.created % 1) * 1e6), os.getpid(), tag,
        os.path.basename(record.pathname), record.lineno, record.getMessage())


def _ensure_handler():
  if not _logger.handlers:
    h = _logging.StreamHandler(sys.stderr)
    h.setFormatter(_Formatter())
    _logger.addHandler(h)
    _logger.setLevel(INFO)


def set_verbosity(level):
  _ensure_handler()
  _logger.setLevel(level)


def get_verbosity():
  return _logger.level


def _log(level, msg, *args):
  _ensure_handler()
  _logger.log(level, msg, *args, stacklevel=3)


def debug(msg, *args):
  _log(DEBUG, msg, *args)


def info(msg, *args):
  _log(INFO, msg, *args)


def warning(msg, *args):
  _log(WARNING, msg, *args)


warn = warning


def error(msg, *args):
  _log(ERROR, msg, *args)


def fatal(msg, *args):
  _log(FATAL, msg, *args)
  raise SystemExit(1)
