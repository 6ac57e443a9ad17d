// la_large.hip -- topics larger than one wave tile.  (stub: filled in next)
#include "la_kernels.h"

namespace la {

hipError_t large_topic_launch(LargeScratch&, const LargeArgs&, bool, hipStream_t) { return hipErrorNotSupported; }

void large_scratch_release(LargeScratch& s) {
    if (s.buf) (void)hipFree(s.buf);
    s.buf = nullptr;
    s.cap = 0;
}

}  // namespace la
