// Host orchestration for CurveBn254
#include "curve_ops_impl.cuh"
#include "prove_impl.cuh"
namespace hk {
extern template struct MsmRun<CurveBn254::Fq>;
extern template struct MsmRun<CurveBn254::Fq2>;
extern template struct MsmSort<CurveBn254::Fr>;
extern template struct PairRun<CurveBn254::Fq::Params>;
const CurveOps* curve_ops_bn254() { return Ops<CurveBn254>::table(); }
}
