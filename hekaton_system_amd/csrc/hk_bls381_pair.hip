// Pairing kernels for CurveBls381 (explicit instantiation; see msm_driver.cuh)
#include "pairing_driver_impl.cuh"
template struct hk::PairRun<hk::Bls381FqP>;
