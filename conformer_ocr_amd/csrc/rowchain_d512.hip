// Translation unit of the row-chain kernels for encoder_dim 512 (rowchain.hip.h).
#include "rowchain.hip.h"

hipError_t launch_rowchain_512(hipStream_t s, const ChainArgs &a, bool taps, int rows_hint) { return launch_rowchain<512>(s, a, taps, rows_hint); }
