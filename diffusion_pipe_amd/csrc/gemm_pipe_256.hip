// gemm_pipe_256.hip -- the 256^2 tile instantiations of the pipelined bf16 GEMM (gemm_pipe_kernel.h: T256S, T256K -- the DiT-sized forward / dgrad problems) as a
// translation unit of their own (build wall time, round 6: see gemm_pipe_group.hip).  Dispatch and planning stay in gemm_pipe.hip.
#include "gemm_pipe_kernel.h"

using namespace dpipe_pipe;

namespace dpipe {

int gemm_pipe_launch_256(int tile, const GemmParams& p, bool a_mc, bool b_mc, int batch, hipStream_t s) {
    return tile == 258 ? launch_pipe<T256K>(p, a_mc, b_mc, batch, s) : launch_pipe<T256S>(p, a_mc, b_mc, batch, s);
}

}  // namespace dpipe
