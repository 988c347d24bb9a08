"""The self-launcher of run_model.py and bench.py (tfrec_amd/launch.py): N child processes with the environment
torch.distributed.run would give them; all children are watched, and a rank that dies ends the others instead of leaving
them at the rendezvous."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tfrec_amd.launch import free_port, launch_ranks  # noqa: E402


def _script(tmp_path, body):
    p = tmp_path / "rank.py"
    p.write_text("import os, sys, time\nrank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])\n" + body)
    return str(p)


def test_every_rank_gets_its_environment(tmp_path):
    s = _script(tmp_path, "open(os.path.join(sys.argv[1], 'r%d' % rank), 'w').write(' '.join(os.environ[k] for k in "
                          "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')))\n")
    assert launch_ranks(3, s, [str(tmp_path)]) == 0
    rows = [open(tmp_path / ("r%d" % r)).read().split() for r in range(3)]
    assert [r[0] for r in rows] == ["0", "1", "2"] and [r[1] for r in rows] == ["0", "1", "2"]
    assert all(r[2] == "3" and r[3] == "3" and r[4] == "127.0.0.1" for r in rows)
    assert len({r[5] for r in rows}) == 1 and int(rows[0][5]) > 0        # one port for all


def test_a_failing_rank_ends_the_others(tmp_path):
    # rank 1 exits with an error at once; ranks 0 and 2 would sleep for a minute (a rendezvous that never completes)
    s = _script(tmp_path, "if rank == 1:\n    sys.exit(7)\ntime.sleep(60)\n")
    t0 = time.time()
    assert launch_ranks(3, s, [], poll_s=0.05) == 7
    assert time.time() - t0 < 20


def test_free_port_is_usable():
    import socket
    p = free_port()
    s = socket.socket()
    s.bind(("127.0.0.1", p))
    s.close()
