"""Writes dataset/movielens-100k-{train,test}.txt (user<TAB>item<TAB>1) from the interaction pairs
kept in tests/golden/ml100k_known_answer.npz, so that run_model.py works out of the box with
tfrec.ini.  (The rating values of the original files are not kept: the path only uses which
(user, item) pairs exist.)"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(ROOT, "tests", "golden", "ml100k_known_answer.npz"))
os.makedirs(os.path.join(ROOT, "dataset"), exist_ok=True)
for name, u, i in (("train", g["train_u"], g["train_i"]), ("test", g["test_u"], g["test_i"])):
    path = os.path.join(ROOT, "dataset", "movielens-100k-%s.txt" % name)
    with open(path, "w") as f:
        for a, b in zip(u.tolist(), i.tolist()):
            f.write("%d\t%d\t1\n" % (a, b))
    print(path, len(u))
