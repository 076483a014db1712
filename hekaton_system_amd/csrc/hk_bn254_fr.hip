// Scalar-field kernels for CurveBn254: digit sort (and, below, NTT / QAP kernels)
#include "msm_driver_impl.cuh"
template struct hk::MsmSort<hk::CurveBn254::Fr>;
