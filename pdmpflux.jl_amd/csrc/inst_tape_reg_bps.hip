// Tape-replay instantiations of the register-form group kernel for sampler S_BPS (inst_reg.hpp with TAPE = true): the
// 13-slot kernels with and without the per-slot skips, every lane count, every built-in separable / banana target, both slope modes.
#include "inst_reg.hpp"

PDMP_NS_BEGIN
hipError_t launch_group_reg_tape_bps(int target, bool fd, const GroupParams &gp, hipStream_t stream) { return reg_by_target<S_BPS, true>(target, fd, gp, stream); }
PDMP_NS_END
