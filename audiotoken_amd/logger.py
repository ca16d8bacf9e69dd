"""stdlib logging helper (reference audiotoken/logger.py:7-31 — out of scope beyond this shim)."""
import logging


def get_logger(name: str, log_file=None, level: str = "ERROR") -> logging.Logger:
    logger = logging.getLogger(name)
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter("%(asctime)s %(name)s %(levelname)s %(message)s"))
        logger.addHandler(h)
    logger.setLevel(getattr(logging, str(level).upper(), logging.ERROR))
    return logger
