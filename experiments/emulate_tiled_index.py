"""Chunk-level emulation of the index logic of experiments/decode4_tiled_gemm.hip (no GPU needed): for every configuration and a set of
ragged shapes, every (weight row-block, m-block) output tile must accumulate every k-block exactly once, and every MFMA must pair a W
chunk and an X chunk of the SAME k-block read from the LDS slot that was filled with exactly that chunk.  The lane-level layouts are
those of the validated dec_gemm (a chunk is the 64-lane x 16-byte operand image of v_mfma_f32_16x16x32_bf16); what is new in the tiled
kernel is only which chunk goes where and when — which is what this checks.    python experiments/emulate_tiled_index.py"""


def emulate(I, J, KG, N, M, K):
    KS = 2; NW = 2 * I; NX = 2 * J; ROWC = NW + NX; CH = ROWC * KS; LPW = CH // 4
    nkb = K // 32; Mb = (M + 15) // 16; nit = (nkb + KS * KG - 1) // (KS * KG)
    done = set()
    for bx in range(N // (32 * I)):
        for by in range((Mb + 2 * J - 1) // (2 * J)):
            rb0, mb0 = bx * NW, by * NX
            acc = {}
            for it in range(nit):
                stage = it & 1
                lds = {}
                for kg in range(KG):
                    for ws in range(4):
                        for u in range(LPW):                       # fetch + park of wave (kg, ws)
                            c = ws + 4 * u; ks = c // ROWC; r = c - ks * ROWC
                            kb = (it * KG + kg) * KS + ks
                            if r < NW:
                                val = ("W", rb0 + r, kb) if kb < nkb else None
                            else:
                                mb = mb0 + (r - NW)
                                val = ("X", mb, kb) if (mb < Mb and kb < nkb) else None
                            lds[(stage, kg, c)] = val
                for kg in range(KG):
                    for ws in range(4):                            # compute of wave (kg, ws)
                        wn, wm = ws & 1, ws >> 1
                        for ks in range(KS):
                            for i in range(I):
                                a = lds[(stage, kg, ks * ROWC + wn * I + i)]
                                for j in range(J):
                                    x = lds[(stage, kg, ks * ROWC + NW + wm * J + j)]
                                    if a is None or x is None:
                                        continue                   # zero chunk: contributes nothing
                                    assert a[0] == "W" and x[0] == "X" and a[2] == x[2], (a, x)
                                    acc.setdefault((ws, i, j), []).append((a[1], x[1], a[2]))      # the K-groups fold into one sum
            for (ws, i, j), lst in acc.items():
                wn, wm = ws & 1, ws >> 1
                rb, mb = rb0 + wn * I + i, mb0 + wm * J + j
                assert all(t[0] == rb and t[1] == mb for t in lst), (rb, mb)
                assert sorted(t[2] for t in lst) == list(range(nkb)), (I, J, KG, rb, mb)
                assert (rb, mb) not in done
                done.add((rb, mb))
    assert done == {(rb, mb) for rb in range(N // 16) for mb in range(Mb)}


if __name__ == "__main__":
    for cfg in [(2, 2, 1), (2, 4, 1), (4, 4, 1), (4, 2, 1), (2, 2, 2), (2, 4, 2), (4, 2, 2)]:
        for (N, M, K) in [(256, 50, 352), (768, 100, 352), (512, 400, 320), (3840, 384, 1280), (1280, 768, 3584)]:
            if N % (32 * cfg[0]) == 0:
                emulate(*cfg, N, M, K)
    print("tiled-GEMM index logic OK")
