// The variable-base MSM on G2: the kernels of msm_impl.h instantiated for the twist (a unit of its own so that the two
// groups compile side by side; see msm.hip for the algorithm).
#include "msm_impl.h"

namespace rlnamd {

struct MsmG2::Impl : MsmImpl<MsmOpsG2> {
  using MsmImpl<MsmOpsG2>::MsmImpl;
};
RLN_MSM_WRAPPERS(MsmG2, MsmOpsG2)

}  // namespace rlnamd
