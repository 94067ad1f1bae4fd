"""Exhaustive check of the rule the host uses to decide that a band cannot clip any anti-diagonal of ksw_extd2/extz2
(ksw2_extd2_sse.c:137-147: st = max(0, r-qlen+1, (r-w+1)>>1), en = min(tlen-1, r, (r+w)>>1)): qlen <= w and tlen <= w + 1."""


def binds(q, t, w):
    for r in range(q + t - 1):
        st, en = max(0, r - q + 1), min(t - 1, r)
        if max(st, (r - w + 1) >> 1) != st or min(en, (r + w) >> 1) != en:
            return True
    return False


if __name__ == "__main__":
    missed = total = 0
    for w in range(1, 48):
        for q in range(1, 100):
            for t in range(1, 100):
                b = binds(q, t, w)
                assert not (b and q <= w and t <= w + 1), (q, t, w)  # the rule is sound
                if not b:
                    total += 1
                    missed += not (q <= w and t <= w + 1)
    print("sound; misses %d of %d non-binding cases" % (missed, total))
