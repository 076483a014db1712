// Host orchestration for CurveBls381
#include "curve_ops_impl.cuh"
#include "prove_impl.cuh"
namespace hk {
extern template struct MsmRun<CurveBls381::Fq>;
extern template struct MsmRun<CurveBls381::Fq2>;
extern template struct MsmSort<CurveBls381::Fr>;
extern template struct PairRun<CurveBls381::Fq::Params>;
const CurveOps* curve_ops_bls381() { return Ops<CurveBls381>::table(); }
}
