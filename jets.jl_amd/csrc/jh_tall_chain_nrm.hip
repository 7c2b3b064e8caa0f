// jh_tall_chain_nrm.hip -- the NORMAL  y = Q(sum_i conj(a_i) .* R(a_i .* P(m))) instantiations of k_chain_adj (jh_tall_chain_kernels.h), a translation unit of their own
// (build time: see jh_tall_chain.hip).
#include "jh_tall_chain_kernels.h"

namespace jhb {
int chain_launch_normal(const jh_chain *ch, void *out, const void *in, int accumulate)
{
    const jh_blockop *op = ch->op;
    const int64_t n = op->row_len[0];
    switch (op->dtype) {
    case JH_F32: return launch_chain_adj<float, 1, 4, 1>(ch, out, in, n, accumulate);
    case JH_F64: return launch_chain_adj<double, 1, 2, 1>(ch, out, in, n, accumulate);
    case JH_C32: return launch_chain_adj<float, 2, 4, 1>(ch, out, in, n * 2, accumulate);
    case JH_C64: return launch_chain_adj<double, 2, 2, 1>(ch, out, in, n * 2, accumulate);
    }
    return jh_fail(JH_ERR_INVALID, "chain_launch_normal: unknown dtype %d", op->dtype);
}
}  // namespace jhb
