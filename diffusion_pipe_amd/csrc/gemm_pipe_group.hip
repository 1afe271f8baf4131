// gemm_pipe_group.hip -- the grouped-launch instantiations of the pipelined bf16 GEMM (gemm_pipe_kernel.h: up to 4 problems of one tile geometry per launch) as a
// translation unit of their own: they are the three largest kernels of the library, and with the rest of gemm_pipe.hip they made one compile that set the build's wall time
// (round 6).  Dispatch and planning stay in gemm_pipe.hip.
#include "gemm_pipe_kernel.h"

using namespace dpipe_pipe;

namespace dpipe {

// geom: 129 = 128^2 on the 2-deep ring (incl. the register-staged members), 128 = 128^2 on the 3-deep ring, else 64^2
int gemm_pipe_launch_group(int geom, const GemmGroup& g, int total_wg, hipStream_t s) {
    switch (geom) {
    case 129: return launch_pipe_group<T128R2>(g, total_wg, s);
    case 128: return launch_pipe_group<T128>(g, total_wg, s);
    default: return launch_pipe_group<T64>(g, total_wg, s);
    }
}

}  // namespace dpipe
