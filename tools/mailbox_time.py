"""Where a single-env step's time goes on the mailbox path: (a) snac_mailbox_step alone through ctypes (doorbell -> resident wave ->
row + acknowledgement), (b) BatchedDMPEnv.mailbox_step, (c) the drop-in class's step(), (d) the launch path (SNAC_MAILBOX=0 form:
step_scalar_wait) -- microseconds per step, median of 5 x `steps`.   usage: mailbox_time.py [kind steps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snac_amd import BatchedDMPEnv  # noqa: E402


def med(f, steps):
    per = []
    for _ in range(5):
        t0 = time.perf_counter()
        f(steps)
        per.append((time.perf_counter() - t0) / steps * 1e6)
    per.sort()
    return per[2], per[0], per[-1]


def main():
    kind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    e = BatchedDMPEnv(kind, True, 1, obs_tail=("record",), seed=1)
    row = e.mailbox_open()
    e.reset_scalar(5, out=row)
    e.sync()
    acts = np.random.RandomState(0).randint(0, 3, steps + 8).tolist()
    step, mb, d, s = e._lib.snac_mailbox_step, e._mb, e._desc_ref, e._state_ref
    e.mailbox_step(0, 1)

    def raw(k):
        for i in range(k):
            step(mb, d, s, acts[i], 1)

    def wrapped(k):
        for i in range(k):
            e.mailbox_step(acts[i], 1)

    print("%dD  raw snac_mailbox_step (ctypes)      %6.2f us  (min %.2f max %.2f)" % ((kind,) + med(raw, steps)))
    print("%dD  BatchedDMPEnv.mailbox_step          %6.2f us  (min %.2f max %.2f)" % ((kind,) + med(wrapped, steps)))
    print("    ", e.mailbox_stats())
    host = e.new_host_obs()

    def launch(k):
        for i in range(k):
            e.step_scalar_wait(acts[i], 1, host)

    launch(10)
    print("%dD  step_scalar_wait (launch + wait)    %6.2f us  (min %.2f max %.2f)" % ((kind,) + med(launch, steps)))
    if kind == 2:
        from snac_amd.envs import deep_mobile_printing_2d1r_dynamic

        for flag in ("1", "0"):
            os.environ["SNAC_MAILBOX"] = flag
            f = deep_mobile_printing_2d1r_dynamic("data_2d_dynamic_dense_envplan_500_train.pkl")
            np.random.seed(1)
            f.reset()

            def cls(k):
                for i in range(k):
                    if f.step(acts[i])[2]:
                        f.reset()

            cls(50)
            print("2D  drop-in class step(), SNAC_MAILBOX=%s  %6.2f us  (min %.2f max %.2f)" % ((flag,) + med(cls, steps)))


def wrapper_times():
    """VectorizedEnvWrapper.step (numpy in, numpy out): the mailbox path (<= 64 envs) against the launch path."""
    from snac_amd.vector import VectorizedEnvWrapper

    table = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "snac_amd", "data", "plans.npz"))["2d_dense_train"]
    for nn in (3, 16, 64):
        for flag in ("1", "0"):
            os.environ["SNAC_MAILBOX"] = flag
            w = VectorizedEnvWrapper((2, True, table), num_envs=nn)
            np.random.seed(1)
            w.reset()
            acts = np.random.RandomState(0).randint(0, 5, (2100, nn))

            def loop(k):
                for i in range(k):
                    w.step(acts[i])

            loop(100)
            m = med(loop, 2000)
            print("2D  VectorizedEnvWrapper.step, %2d envs, SNAC_MAILBOX=%s  %6.2f us per vector step  = %.3e env-steps/s" % (nn, flag, m[0], nn / m[0] * 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "wrapper":
        wrapper_times()
        sys.exit(0)
    main()
