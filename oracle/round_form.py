"""An independent numpy checker built on the round property of the greedy (SURVEY.md section 8a note 5); shared by
the CPU test that validates it against the literal oracle and by the full-size GPU tests that are too large for the
literal form."""
import numpy as np


def round_form(part_off, pid, lag, cons_off, ranks):
    """Independent checker built on the round property (SURVEY.md section 8a note 5): sort by (lag desc, id asc);
    in each round of C positions the k-th partition goes to the k-th consumer by (total, rank) at round start."""
    exp_p = np.empty_like(pid)
    exp_m = np.empty_like(pid)
    exp_t = np.zeros(ranks.size, dtype=np.int64)
    for t in range(part_off.size - 1):
        a, z = int(part_off[t]), int(part_off[t + 1])
        ca, cz = int(cons_off[t]), int(cons_off[t + 1])
        order = np.lexsort((pid[a:z], ~lag[a:z]))              # ~x = -x-1: descending lag without overflow
        sp, sl = pid[a:z][order], lag[a:z][order]
        exp_p[a:z] = sp
        c = cz - ca
        if c == 0:
            exp_m[a:z] = -1
            continue
        tot = np.zeros(c, dtype=np.int64)
        r = ranks[ca:cz]
        for lo in range(0, z - a, c):
            k = min(c, z - a - lo)
            who = np.lexsort((r, tot))[:k]
            exp_m[a + lo:a + lo + k] = r[who]
            with np.errstate(over="ignore"):
                tot[who] += sl[lo:lo + k]
        exp_t[ca:cz] = tot
    return exp_p, exp_m, exp_t
