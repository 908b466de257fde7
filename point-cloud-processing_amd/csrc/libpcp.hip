// Unity translation unit of libpcp_hip.so (one code object, kernels shared between parts).
#include "stage1.hip"
#include "crop.hip"
#include "fps.hip"
#include "misc.hip"
#include "voxel.hip"
#include "sor.hip"
