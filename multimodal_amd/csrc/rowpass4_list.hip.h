// Every instantiation of k_rowpass4 the library launches (launch_rowpass4_kt in api_loop.hip), as X-macro lists: the
// translation units rowpass4_inst_*.hip instantiate them (explicit instantiation definitions), ctx.hip.h only declares them
// (extern template), so that the instantiations compile in parallel with the host-side units (__graft_entry__.compile_library).
//   X(KT, ODD, MODE, EP, NW, SPLIT, Q8)
#pragma once

// k <= 224: 8-wave workgroups.  Update pass: whole rows / column-split, each with 16-bit tiles, fp8 tiles, fp8 tiles without
// the numerator's eps; W0 = V.H0^T and loss-only passes: whole rows.
#define KL_RP4_SMALL_OE(X, KT, ODD, EP)                                                                          \
    X(KT, ODD, 0, EP, 8, 0, 0) X(KT, ODD, 0, EP, 8, 1, 0) X(KT, ODD, 0, EP, 8, 0, 1) X(KT, ODD, 0, EP, 8, 1, 1)  \
    X(KT, ODD, 0, EP, 8, 0, 2) X(KT, ODD, 0, EP, 8, 1, 2) X(KT, ODD, 1, EP, 8, 0, 0) X(KT, ODD, 2, EP, 8, 0, 0)
#define KL_RP4_SMALL(X, KT) KL_RP4_SMALL_OE(X, KT, 0, 0) KL_RP4_SMALL_OE(X, KT, 0, 1) KL_RP4_SMALL_OE(X, KT, 1, 0) KL_RP4_SMALL_OE(X, KT, 1, 1)
// 224 < k <= 512: 4-wave workgroups (FUSED order), even KT only
#define KL_RP4_BIG_E(X, KT, EP) X(KT, 0, 0, EP, 4, 0, 0) X(KT, 0, 0, EP, 4, 0, 1) X(KT, 0, 1, EP, 4, 0, 0) X(KT, 0, 2, EP, 4, 0, 0)
#define KL_RP4_BIG(X, KT) KL_RP4_BIG_E(X, KT, 0) KL_RP4_BIG_E(X, KT, 1)

#define KL_RP4_LIST_1(X) KL_RP4_SMALL(X, 1) KL_RP4_SMALL(X, 2) KL_RP4_SMALL(X, 3) KL_RP4_SMALL(X, 4)
#define KL_RP4_LIST_2(X) KL_RP4_SMALL(X, 5) KL_RP4_SMALL(X, 6) KL_RP4_SMALL(X, 7)
#define KL_RP4_LIST_3(X) KL_RP4_BIG(X, 8) KL_RP4_BIG(X, 10) KL_RP4_BIG(X, 12) KL_RP4_BIG(X, 14) KL_RP4_BIG(X, 16)

#define KL_RP4_DEFINE(KT, ODD, MODE, EP, NW, SPLIT, Q8) template __global__ void k_rowpass4<KT, ODD, MODE, EP, NW, SPLIT, Q8>(RowPass4Args);
#define KL_RP4_DECLARE(KT, ODD, MODE, EP, NW, SPLIT, Q8) extern template __global__ void k_rowpass4<KT, ODD, MODE, EP, NW, SPLIT, Q8>(RowPass4Args);
