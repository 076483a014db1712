// Pairing kernels for CurveBn254 (explicit instantiation; see msm_driver.cuh)
#include "pairing_driver_impl.cuh"
template struct hk::PairRun<hk::Bn254FqP>;
