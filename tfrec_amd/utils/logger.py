"""Logger with the reference's interface (utils/logger.py:11-44): a logger named "TFRec" that
writes every message to a file and to stdout with a timestamp."""
import logging
import sys


class Logger(object):
    def __init__(self, filename):
        self.logger = logging.getLogger("TFRec")
        self.logger.setLevel(logging.DEBUG)
        fmt = logging.Formatter("%(asctime)s - %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
        for handler in (logging.FileHandler(filename), logging.StreamHandler(sys.stdout)):
            handler.setLevel(logging.DEBUG)
            handler.setFormatter(fmt)
            self.logger.addHandler(handler)

    def debug(self, message):
        self.logger.debug(message)

    def info(self, message):
        self.logger.info(message)

    def warning(self, message):
        self.logger.warning(message)

    def error(self, message):
        self.logger.error(message)

    def critical(self, message):
        self.logger.critical(message)
