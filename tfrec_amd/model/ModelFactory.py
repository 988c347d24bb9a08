"""Configuration reading and model selection (model/ModelFactory.py:15-89).

Same files and keys as the reference: ``tfrec.ini`` section [tfrec] (data_name, train_file,
test_file, format, separator, model) and ``conf/<model>.ini`` section [hyperparameters].  The
reference passes every value through Python's eval(); here values are parsed as literals and
plain arithmetic only ("0.1/16", "'\\t'", "[1, 2]"), anything else stays a string."""
import ast
import configparser
import operator
import os
import time
from collections import OrderedDict

from ..utils import Logger
from .APR import APR
from .BPR import BPR
from .DNSBPR import DNSBPR
from .Pop import Pop

_BIN = {ast.Add: operator.add, ast.Sub: operator.sub, ast.Mult: operator.mul, ast.Div: operator.truediv,
        ast.FloorDiv: operator.floordiv, ast.Pow: operator.pow, ast.Mod: operator.mod}
_UN = {ast.UAdd: operator.pos, ast.USub: operator.neg}


def parse_value(text):
    """Literal / arithmetic evaluation of one ini value; the raw string if it is neither."""
    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.Constant):
            return node.value
        if isinstance(node, ast.BinOp) and type(node.op) in _BIN:
            return _BIN[type(node.op)](ev(node.left), ev(node.right))
        if isinstance(node, ast.UnaryOp) and type(node.op) in _UN:
            return _UN[type(node.op)](ev(node.operand))
        if isinstance(node, (ast.List, ast.Tuple)):
            vals = [ev(e) for e in node.elts]
            return vals if isinstance(node, ast.List) else tuple(vals)
        if isinstance(node, ast.Dict):
            return {ev(k): ev(v) for k, v in zip(node.keys, node.values)}
        raise ValueError("not a literal")
    try:
        return ev(ast.parse(text.strip(), mode="eval"))
    except Exception:
        return text


class _NoLog:
    def info(self, *a, **k):
        pass

    debug = warning = error = critical = info


class ModelFactory(object):
    def __init__(self, root="."):
        self.root = root
        self.model_dict = {"BPR": BPR, "DNSBPR": DNSBPR, "APR": APR, "Pop": Pop}

    def options(self):
        """The merged, parsed configuration without any side effect (no log file is opened)."""
        return self._eval_parameter(*self._read_config())

    def devices(self):
        """GPUs the run asks for (``devices``, optional, in either ini file; default 1: the reference's single process)."""
        return int(self.options().get("devices", 1))

    def get_model(self):
        tfrec_config, model_config = self._read_config()
        model_name = tfrec_config["model"]
        if model_name not in self.model_dict:
            raise KeyError("model '%s' is not part of this package (available: %s)" % (model_name, ", ".join(sorted(self.model_dict))))
        Model = self.model_dict[model_name]
        # one log file per run: in a run of several processes (one per GPU) rank 0 writes it
        Model.logger = self._create_logger(tfrec_config, model_config) if int(os.environ.get("RANK", "0")) == 0 else _NoLog()
        return Model, self._eval_parameter(tfrec_config, model_config)

    def _read_config(self):
        config = configparser.ConfigParser()
        if not config.read(os.path.join(self.root, "tfrec.ini")):
            raise FileNotFoundError("tfrec.ini not found under %s" % os.path.abspath(self.root))
        tfrec_config = OrderedDict(config["tfrec"].items())
        model_config_path = os.path.join(self.root, "conf", tfrec_config["model"] + ".ini")
        mconf = configparser.ConfigParser()
        if not mconf.read(model_config_path):
            raise FileNotFoundError(model_config_path)
        return tfrec_config, OrderedDict(mconf["hyperparameters"].items())

    def _create_logger(self, tfrec_config, model_config):
        data_name, model_name = tfrec_config["data_name"], tfrec_config["model"]
        log_dir = os.path.join(self.root, "log", data_name, model_name)
        os.makedirs(log_dir, exist_ok=True)
        # file name: model, hyperparameters (short values only), timestamp -- as the reference builds it
        tag = '_'.join("{}={}".format(k, v) for k, v in model_config.items() if len(v) < 20)
        tag = ''.join('_' if c in '/\\":*?<>|\t' else c for c in tag)
        logger = Logger(os.path.join(log_dir, "%s_log_%s_%d.log" % (model_name, tag, time.time())))
        logger.info("\nTFRec information:\n%s " % '\n'.join("{}={}".format(k, v) for k, v in tfrec_config.items()))
        logger.info("\n")
        logger.info("Recommender:%s" % model_name)
        logger.info("Dataset name:\t%s" % data_name)
        logger.info("\nHyperparameters:\n%s " % '\n'.join("{}={}".format(k, v) for k, v in model_config.items()))
        return logger

    def _eval_parameter(self, tfrec_config, model_config):
        config = OrderedDict(tfrec_config, **model_config)
        for key, value in config.items():
            config[key] = parse_value(value)
        return config
