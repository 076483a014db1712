// G2 bucket kernels for CurveBls381 (explicit instantiation; see msm_driver.cuh)
#include "msm_driver_impl.cuh"
template struct hk::MsmRun<hk::CurveBls381::Fq2>;
