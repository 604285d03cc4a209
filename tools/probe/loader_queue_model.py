"""Model of one loader wave's in-order DMA queue in gemm144l_dma_kernel built with -DPRIMX_G144L_XLDS=1 (csrc/gemm.hip): checks that
the `s_waitcnt vmcnt(N)` in front of every barrier implies the data the barrier publishes, for every K / 64 >= 2, and that no more
than 63 instructions are ever outstanding (the counter's range).  Host-only; the kernel itself is untested (DESIGN_LOG.md section 9)."""
NL = 17   # DMA wave-instructions per tile and loader wave


def run(nk):
    q, landed = [], set()

    def issue(name, n):
        q.extend((name, i) for i in range(n))

    def wait(n):   # vmcnt(n): everything but the newest n has landed
        landed.update(q[:max(0, len(q) - n)])

    def tile(k):
        return f"tile{min(k, nk - 1)}@{k}"   # clamped re-fetches are distinct queue entries

    def has(name, n=NL):
        return all((name, i) in landed for i in range(n))

    worst = 0
    issue(tile(0), NL); issue(tile(1), NL); issue(tile(2), NL)
    wait(2 * NL)
    assert has(tile(0))                                   # P
    issue("XA", 7)
    for kt in range(nk):
        worst = max(worst, len([x for x in q if x not in landed]))
        wait({0: NL + 7, 1: NL + 14, 2: NL + 13, 3: NL + 6}.get(kt, NL))
        assert has(tile(kt + 1)), (nk, kt)                # S_kt
        issue(tile(kt + 3), NL)
        if kt == 0:
            issue("XB", 7)
        if kt == 1:
            issue("XC", 6)
        worst = max(worst, len([x for x in q if x not in landed]))
    wait(0)                                               # D
    assert has("XA", 7) and has("XB", 7) and has("XC", 6) and all(x in landed for x in q)
    assert worst <= 63
    return worst


if __name__ == "__main__":
    for nk in range(2, 80):
        run(nk)
    print("ok: K / 64 = 2 .. 79, at most", max(run(n) for n in range(2, 80)), "DMA instructions outstanding per loader wave")
