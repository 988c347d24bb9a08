"""Entry point with the reference's shape (split_data.py:1-10): reads the [split] section of
split_data.ini and writes <data_file>.train / .test / .user2id / .item2id.  The generator is seeded so
that a run is repeatable (the reference leaves it unseeded)."""
import configparser

import numpy as np

from tfrec_amd.data import Splitter

if __name__ == "__main__":
    config = configparser.ConfigParser()
    if not config.read("split_data.ini"):
        raise FileNotFoundError("split_data.ini")
    np.random.seed(2018)
    Splitter(dict(config.items("split"))).split()
