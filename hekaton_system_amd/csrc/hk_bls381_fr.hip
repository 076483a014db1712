// Scalar-field kernels for CurveBls381: digit sort (and, below, NTT / QAP kernels)
#include "msm_driver_impl.cuh"
template struct hk::MsmSort<hk::CurveBls381::Fr>;
