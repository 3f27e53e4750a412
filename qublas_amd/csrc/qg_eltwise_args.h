// qg_eltwise_args.h — plain argument structs of the element-wise epilogue (host and device code)
#pragma once
#include "qg_plan.h"

// run-time operands of the stages (device pointers to packed tensors in the plan's packed-C index space, or scalars)
struct QEpArgs {
    const char* e[QG_MAX_EW];
    int64_t scalar[QG_MAX_EW];
};

// arguments of the stand-alone pass: packed C (cbytes containers, n elements incl. padding) -> packed D
struct QEltwiseArgs {
    const char* C;
    char* D;
    int64_t n;
    int32_t cbytes, pad_;
    QEpTable t;
    QEpArgs a;
};

