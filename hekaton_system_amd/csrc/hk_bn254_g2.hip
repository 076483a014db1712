// G2 bucket kernels for CurveBn254 (explicit instantiation; see msm_driver.cuh)
#include "msm_driver_impl.cuh"
template struct hk::MsmRun<hk::CurveBn254::Fq2>;
