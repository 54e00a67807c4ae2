// Operation layer: OperationType, the OperationTraits primary template, and the CDNA4 (Rocm) op
// classes it resolves to.  Mirrors /root/reference/Mila/Src/Dnn/Compute/Operations/
// OperationTraits.Template.ixx:59-60,78-79,95-147, OperationType.ixx:29-49, OperationBase.ixx:21-175
// and is the build's counterpart of Compute/Devices/Cuda/Operations/OperationTraits.Cuda.ixx.
//
// Contract kept from the reference: ops are constructed by the component as
// std::make_shared<OpType>(IExecutionContext*, const Config&); they cache RAW pointers handed over by
// setParameters()/setWeightScales() and never own parameters; build() validates shapes and throws;
// forward() only enqueues work on the context's stream; scratch is fetched from the context on every
// forward.  Every forward() ends in exactly one C-ABI call (include/mila_cdna4.h).
#pragma once

#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "Core.h"
#include "Quantization.h"

namespace Mila::Dnn::Compute
{
    enum class OperationType
    {
        LinearOp, GroupedQueryAttentionOp, MultiHeadAttentionOp, RmsNormOp, LayerNormOp, SoftmaxOp, GeluOp,
        GegluOp, SwigluOp, RopeOp, LpeOp, TokenEmbeddingOp, ResidualOp, SamplingOp,
    };

    /// primary stays undefined: a missing (Op, Device, Precision, Policy) row is a compile error
    template<OperationType TOp, DeviceType TDeviceType, TensorDataType TPrecision, typename TPolicy = void>
    struct OperationTraits;

    template<OperationType TOp, DeviceType TDeviceType, TensorDataType TPrecision, typename TPolicy = void>
    concept OperationSupported = requires { sizeof( OperationTraits<TOp, TDeviceType, TPrecision, TPolicy> ); };

    template<typename TOp, typename TTensor>
    concept UnaryOpConcept = requires( const TOp& op, const TTensor& in, TTensor& out ) { op.forward( in, out ); };

    template<typename TOp, typename TTensor>
    concept LinearOpConcept = requires( const TOp& op, const TTensor& in, TTensor& out ) { op.forward( in, out ); };

    /// Operation<Dev,Prec> base (OperationBase.ixx:21-175): holds the typed context
    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class Operation
    {
    public:
        explicit Operation( IExecutionContext* ctx ) : context_( cast_context<TDeviceType>( ctx ) ) {}
        virtual ~Operation() = default;
        ExecutionContext<TDeviceType>* getExecutionContext() const noexcept { return context_; }
    protected:
        ExecutionContext<TDeviceType>* context_;
    };

    using RocmBf16Tensor = Tensor<TensorDataType::BF16, RocmDeviceMemoryResource>;
    template<TensorDataType TPrecision> using RocmTensor = Tensor<TPrecision, RocmDeviceMemoryResource>;
    /// the precisions the CDNA4 op rows exist for: BF16 (the product path) and FP32 (the reference's "validation and reference" rows, OperationTraits.Cuda.ixx:50-54,
    /// :108-126, :274-282 -- Linear, LayerNorm, Softmax, GELU, Residual, MHA, LPE, RoPE, RMSNorm; csrc/fp32_rows.hip)
    template<TensorDataType TPrecision> concept RocmPrecision = TPrecision == TensorDataType::BF16 || TPrecision == TensorDataType::FP32;

    // ---------------------------------------------------------------------------------------
    // Linear
    // ---------------------------------------------------------------------------------------
    struct LinearOpConfig
    {
        dim_t in_features{ 0 }, out_features{ 0 };
        bool has_bias{ false };
    };

    /// Test instrument (off unless a test installs one): records the per-token e4m3 activations a Linear on an fp8 x fp8 prefill path (W4A8, W8A8) consumed -- the bytes
    /// and the scales, copied to the host behind a stream synchronize.  A teacher-forced oracle then multiplies THOSE bytes layer by layer, so that a whole-model logit
    /// comparison is not at the mercy of e4m3 code flips upstream (tests/test_gemma_conditioned_gpu.py).  One pointer test per forward when off.
    struct ActivationTap
    {
        struct Record { int M, K, N; std::vector<uint8_t> x8; std::vector<float> ts; };
        std::vector<Record> records;
    };
    inline ActivationTap*& activationTap() { static ActivationTap* tap = nullptr; return tap; }

    /// counterpart of CudaLinearOp<Prec, TWeightQuant> (OPS/Linear/CudaLinearOp.ixx:107-1287)
    template<TensorDataType TPrecision, Quant::Weight::WeightQuantPolicy TWeightQuant>
    class RocmLinearOp : public Operation<DeviceType::Rocm, TPrecision>
    {
        static_assert( TPrecision == TensorDataType::BF16 || ( TPrecision == TensorDataType::FP32 && !TWeightQuant::kIsQuantized ),
                       "the CDNA4 Linear rows: BF16 activations under every weight policy, FP32 with unquantized weights (the validation row)" );
    public:
        using TensorType = RocmTensor<TPrecision>;
        static constexpr bool kFp32 = TPrecision == TensorDataType::FP32;
        static constexpr int kFmt = Quant::Weight::abiWeightFormat<TWeightQuant>();
        static constexpr int kGroup = Quant::Weight::groupSizeOf<TWeightQuant>();

        RocmLinearOp( IExecutionContext* ctx, const LinearOpConfig& cfg ) : Operation<DeviceType::Rocm, TPrecision>( ctx ), cfg_( cfg )
        {
            if ( cfg.in_features <= 0 || cfg.out_features <= 0 ) throw std::invalid_argument( "RocmLinearOp: feature counts must be positive" );
            if constexpr ( kFmt == 2 )
                if ( cfg.in_features % kGroup != 0 ) throw std::invalid_argument( "RocmLinearOp: in_features must be a multiple of the FP4 group size" );
        }

        /// caches raw pointers; never owns (OperationBase.ixx:52-64, CudaLinearOp.ixx:192-241)
        void setParameters( ITensor* weight, ITensor* bias )
        {
            if ( !weight ) throw std::invalid_argument( "RocmLinearOp::setParameters: weight is required" );
            if ( cfg_.has_bias && !bias ) throw std::invalid_argument( "RocmLinearOp::setParameters: bias is required by the config" );
            weight_ = weight->rawData();
            bias_ = bias ? static_cast<const uint16_t*>( bias->rawData() ) : nullptr;      // (FP32 row: float elements behind the same pointer)
        }

        void setWeightScales( ITensor* scales )
        {
            if constexpr ( !TWeightQuant::kIsQuantized ) throw std::logic_error( "RocmLinearOp::setWeightScales: policy is not quantized" );
            if ( !scales ) throw std::invalid_argument( "RocmLinearOp::setWeightScales: null scales" );
            scales_ = static_cast<const float*>( scales->rawData() );
        }

        void build( const BuildContext& ctx )
        {
            const auto& s = ctx.inputShape();
            if ( s.back() != cfg_.in_features )
                throw std::invalid_argument( "RocmLinearOp::build: input feature dimension " + std::to_string( s.back() ) +
                                             " does not match in_features " + std::to_string( cfg_.in_features ) );
            if ( !weight_ ) throw std::runtime_error( "RocmLinearOp::build: setParameters() must be called first" );
            if ( TWeightQuant::kIsQuantized && !scales_ ) throw std::runtime_error( "RocmLinearOp::build: setWeightScales() must be called first" );
            built_ = true;
        }

        /// M == 1 -> decode matvec; M > 1 -> MFMA GEMM (CudaLinearOp.ixx:535-827)
        void forward( const TensorType& in, TensorType& out ) const
        {
            if ( !built_ ) throw std::runtime_error( "RocmLinearOp::forward: not built" );
            const int K = narrowToKernelIndex( cfg_.in_features, "in_features" );
            const int N = narrowToKernelIndex( cfg_.out_features, "out_features" );
            const int M = narrowToKernelIndex( static_cast<dim_t>( in.size() ) / cfg_.in_features, "outer size" );
            mila_stream_t st = this->context_->getStream();
            if constexpr ( kFp32 )
            {
                // the FP32 row (CudaMatVecBias.Fp32.cu:40, CudaMatMulFp32.cu:31-183): fp32 in / accumulate / out
                auto* yf = static_cast<float*>( out.rawData() );
                auto* xf = static_cast<const float*>( in.rawData() );
                const auto* wf = static_cast<const float*>( weight_ );
                const auto* bf = reinterpret_cast<const float*>( bias_ );
                if ( M == 1 ) rocmCheck( mila_cdna4_matvec_fp32( yf, xf, wf, bf, K, N, st ) );
                else rocmCheck( mila_cdna4_gemm_fp32( yf, xf, wf, bf, M, K, N, 0, st ) );
                return;
            }
            auto* y = static_cast<uint16_t*>( out.rawData() );
            auto* x = static_cast<const uint16_t*>( in.rawData() );
            if ( M == 1 )
            {
                if constexpr ( kFmt == 0 ) rocmCheck( mila_cdna4_matvec_bf16( y, x, static_cast<const uint16_t*>( weight_ ), bias_, K, N, st ) );
                else if constexpr ( kFmt == 1 ) rocmCheck( mila_cdna4_matvec_bf16_qfp8( y, x, static_cast<const uint8_t*>( weight_ ), scales_, bias_, K, N, st ) );
                else rocmCheck( mila_cdna4_matvec_bf16_qfp4( y, x, static_cast<const uint8_t*>( weight_ ), scales_, bias_, K, N, kGroup, st ) );
            }
            else
            {
                if constexpr ( kFmt == 0 ) gemmWithWorkspace( y, x, static_cast<const uint16_t*>( weight_ ), M, K, N, 0, st );
                else
                {
                    if constexpr ( kFmt == 1 )
                    {
                        // W8A8 (opt-in, setFp8ActivationPrefill): the policy's own e4m3 weights + per-channel scales on the fp8 matrix cores, per-token e4m3 activations --
                        // "FP8 matmul consumes weights and scales natively" (Quantization/Weight/Policies.ixx:39-40); no staging pass, no bf16 copy of the weights
                        if ( use_fp8_activation_prefill_ && mila_cdna4_gemm_fp8_applicable( M, K, N ) )
                        {
                            uint8_t* x8; float* ts; void* ws;
                            const size_t ws_bytes = mila_cdna4_gemm_fp8_workspace_bytes( M, K, N );
                            activationScratch( M, K, x8, ts, ws_bytes, &ws );
                            rocmCheck( mila_cdna4_quantize_fp8_per_token( x8, ts, x, M, K, st ) );
                            recordTap( x8, ts, M, K, N );
                            rocmCheck( mila_cdna4_gemm_fp8_w8a8_ws( y, x8, static_cast<const uint8_t*>( weight_ ), ts, scales_, bias_, M, K, N, ws, ws_bytes, st ) );
                            return;
                        }
                        // resident prefill weights: the staging pass was run once, at load (same values => same bits as the staged call)
                        if ( resident_bf16_ && mila_cdna4_gemm_staging_bytes( M, K, N ) != 0 )
                        {
                            gemmWithWorkspace( y, x, resident_bf16_->data(), M, K, N, 0, st );
                            return;
                        }
                    }
                    // 2-phase staging through context scratch when the LDS-DMA GEMM applies; scratch is fetched per
                    // forward and never cached (reference rule, CudaLinearOp.ixx:603-614)
                    if constexpr ( kFmt == 2 )
                    {
                        // W4A8: fp4 -> e4m3 weight staging + per-token e4m3 activations + fp8 x fp8 MFMA GEMM, the reference's default
                        // prefill for this policy (kUseFp8ActivationPrefillPath, CudaLinearOp.ixx:646-715), when an fp8 kernel serves the shape
                        if ( use_fp8_activation_prefill_ && weight_fp8_scale_ && resident_e4m3_ && mila_cdna4_gemm_fp8_applicable( M, K, N ) )
                        {
                            uint8_t* x8; float* ts; void* ws;
                            const size_t ws_bytes = mila_cdna4_gemm_fp8_workspace_bytes( M, K, N );
                            activationScratch( M, K, x8, ts, ws_bytes, &ws );
                            rocmCheck( mila_cdna4_quantize_fp8_per_token( x8, ts, x, M, K, st ) );
                            recordTap( x8, ts, M, K, N );
                            rocmCheck( mila_cdna4_gemm_fp8_scaled_ws( y, x8, resident_e4m3_->data(), ts, weight_fp8_scale_->data(), bias_, M, K, N, ws, ws_bytes, st ) );
                            return;
                        }
                        if ( use_fp8_activation_prefill_ && weight_fp8_scale_ && mila_cdna4_gemm_fp8_applicable( M, K, N ) )
                        {
                            const size_t need8 = mila_cdna4_gemm_w4a8_scratch_bytes( M, K, N );
                            void* scratch8 = this->context_->getScratch( need8 );
                            rocmCheck( mila_cdna4_gemm_bf16_w4a8( y, x, static_cast<const uint8_t*>( weight_ ), scales_, weight_fp8_scale_->data(), bias_, M, K, N, kGroup, scratch8, need8, st ) );
                            return;
                        }
                    }
                    const size_t need = mila_cdna4_gemm_staging_bytes( M, K, N );
                    void* scratch = need ? this->context_->getScratch( need ) : nullptr;
                    if constexpr ( kFmt == 1 ) rocmCheck( mila_cdna4_gemm_bf16_w8a16_staged( y, x, static_cast<const uint8_t*>( weight_ ), scales_, bias_, M, K, N, scratch, need, st ) );
                    else rocmCheck( mila_cdna4_gemm_bf16_w4a16_staged( y, x, static_cast<const uint8_t*>( weight_ ), scales_, bias_, M, K, N, kGroup, scratch, need, st ) );
                }
            }
        }

        /// true when forwardGelu() serves this op and row count: unquantized weights on the GEMM branch
        bool fusesGelu( dim_t rows ) const noexcept { return kFmt == 0 && rows > 1; }

        /// Linear -> tanh-GELU in one kernel (the MLP's fc_1 -> gelu, MLP.ixx:148-161): the bits of forward() followed by RocmGeluOp::forward()
        void forwardGelu( const TensorType& in, TensorType& out ) const
        {
            if ( !built_ ) throw std::runtime_error( "RocmLinearOp::forwardGelu: not built" );
            const int K = narrowToKernelIndex( cfg_.in_features, "in_features" );
            const int N = narrowToKernelIndex( cfg_.out_features, "out_features" );
            const int M = narrowToKernelIndex( static_cast<dim_t>( in.size() ) / cfg_.in_features, "outer size" );
            if ( !fusesGelu( M ) ) throw std::logic_error( "RocmLinearOp::forwardGelu: only unquantized weights at more than one row" );
            if constexpr ( kFp32 )
                rocmCheck( mila_cdna4_gemm_fp32( static_cast<float*>( out.rawData() ), static_cast<const float*>( in.rawData() ), static_cast<const float*>( weight_ ),
                                                 reinterpret_cast<const float*>( bias_ ), M, K, N, 1, this->context_->getStream() ) );
            else if constexpr ( kFmt == 0 )
                gemmWithWorkspace( static_cast<uint16_t*>( out.rawData() ), static_cast<const uint16_t*>( in.rawData() ), static_cast<const uint16_t*>( weight_ ), M, K, N, 1,
                                   this->context_->getStream() );
        }

        /// the bf16 GEMM with the context's scratch as its workspace, as CudaLinearOp hands context_->getCublasLtWorkspace() to every plan (CudaLinearOp.ixx:637-638,
        /// :817-818): short prompts and remainders split K through it.  Fetched per forward, never cached.
        void gemmWithWorkspace( uint16_t* y, const uint16_t* x, const uint16_t* w, int M, int K, int N, int act, mila_stream_t st ) const
        {
            const size_t need = mila_cdna4_gemm_workspace_bytes( M, K, N );
            void* ws = need ? this->context_->getScratch( need ) : nullptr;
            rocmCheck( mila_cdna4_gemm_bf16_ws( y, x, w, bias_, M, K, N, act, ws, need, st ) );
        }

        void backward( const TensorType&, const TensorType&, TensorType& ) const
        {
            throw std::logic_error( "RocmLinearOp::backward: the CDNA4 backend is inference-only (and the reference forbids backward on quantized weights)" );
        }

        /// quantize-on-load: bf16 source already on the device (CudaLinearOp.ixx:332-392)
        void quantize( const uint16_t* src_bf16_device, ITensor& weight_out, ITensor& scales_out ) const
        {
            const int K = narrowToKernelIndex( cfg_.in_features, "in_features" );
            const int N = narrowToKernelIndex( cfg_.out_features, "out_features" );
            mila_stream_t st = this->context_->getStream();
            if constexpr ( kFmt == 1 )
                rocmCheck( mila_cdna4_quantize_fp8_per_channel( static_cast<uint8_t*>( weight_out.rawData() ), static_cast<float*>( scales_out.rawData() ), src_bf16_device, N, K, st ) );
            else if constexpr ( kFmt == 2 )
                rocmCheck( mila_cdna4_quantize_fp4_per_group( static_cast<uint8_t*>( weight_out.rawData() ), static_cast<float*>( scales_out.rawData() ), src_bf16_device, N, K, kGroup, st ) );
            else
                throw std::logic_error( "RocmLinearOp::quantize: NoWeightQuant has no quantize path" );
        }

        /// fp4 policy: (re)compute the per-tensor e4m3 weight scale the W4A8 prefill needs (an op-owned device scalar, like
        /// CudaLinearOp's weight_fp8_scale_, CudaLinearOp.ixx:311-330, :1118-1129)
        void onQuantizedWeightsLoaded()
        {
            if constexpr ( kFmt == 2 )
            {
                if ( !scales_ ) return;
                if ( !weight_fp8_scale_ ) weight_fp8_scale_ = std::make_unique<Tensor<TensorDataType::FP32, RocmDeviceMemoryResource>>( this->context_->getDeviceId(), shape_t{ 1 } );
                rocmCheck( mila_cdna4_fp4_weight_fp8_scale( weight_fp8_scale_->data(), scales_, (int64_t)cfg_.out_features * ( cfg_.in_features / kGroup ), this->context_->getStream() ) );
            }
            refreshResident();
        }
        /// Resident prefill weights (default on): the staging pass of the quantized prefill -- fp8 -> bf16 (W8A16), fp4 -> e4m3 (W4A8) --
        /// is run ONCE when the weights are loaded and kept in an op-owned tensor (+2 / +1 bytes per weight), instead of into context
        /// scratch on every forward as the reference must on a 12 GB card (CudaLinearOp.ixx:597-644, :646-715).  Decode still streams the
        /// quantized weights; prefill results are bit-identical either way.
        void setResidentPrefillWeights( bool on )
        {
            resident_ = on;
            if ( !on ) { resident_bf16_.reset(); resident_e4m3_.reset(); }
            else refreshResident();
        }
        bool residentPrefillWeights() const noexcept { return resident_; }
        const uint16_t* residentBf16() const noexcept { return resident_bf16_ ? resident_bf16_->data() : nullptr; }
        const uint8_t* residentE4m3() const noexcept { return resident_e4m3_ ? resident_e4m3_->data() : nullptr; }
        /// true when forward() of M rows would take the resident W4A8 path (fp8 x fp8 GEMM on weights staged at load): a producer may then hand over its
        /// output already quantized per token (forwardFp8Activations) instead of as bf16
        bool acceptsFp8Activations( int M ) const
        {
            if constexpr ( kFmt == 0 ) return false;
            else if constexpr ( kFmt == 1 ) return use_fp8_activation_prefill_ && mila_cdna4_gemm_fp8_applicable( M, (int)cfg_.in_features, (int)cfg_.out_features ) != 0;
            else return use_fp8_activation_prefill_ && weight_fp8_scale_ && resident_e4m3_ && mila_cdna4_gemm_fp8_applicable( M, (int)cfg_.in_features, (int)cfg_.out_features ) != 0;
        }
        /// the W4A8 forward on activations the caller quantized (x8 [M, K] e4m3, ts [M] per-token scales -- exactly what quantize_fp8_per_token gives): the same
        /// GEMM call forward() makes, so the output carries the same bits
        /// (x8 / ts are the caller's own buffers, NOT context scratch: the GEMM's split-K workspace is taken from there)
        void forwardFp8Activations( const uint8_t* x8, const float* ts, uint16_t* y, int M )
        {
            if ( !acceptsFp8Activations( M ) ) throw std::logic_error( "RocmLinearOp::forwardFp8Activations: the fp8 x fp8 path does not serve this call" );
            const int K = (int)cfg_.in_features, N = (int)cfg_.out_features;
            const size_t ws_bytes = mila_cdna4_gemm_fp8_workspace_bytes( M, K, N );
            void* ws = ws_bytes ? this->context_->getScratch( ws_bytes ) : nullptr;
            if constexpr ( kFmt == 1 )
                rocmCheck( mila_cdna4_gemm_fp8_w8a8_ws( y, x8, static_cast<const uint8_t*>( weight_ ), ts, scales_, bias_, M, K, N, ws, ws_bytes, this->context_->getStream() ) );
            else
                rocmCheck( mila_cdna4_gemm_fp8_scaled_ws( y, x8, resident_e4m3_->data(), ts, weight_fp8_scale_->data(), bias_, M, K, N, ws, ws_bytes, this->context_->getStream() ) );
        }
        /// the test instrument above: copy this call's e4m3 activations and scales to the host when a tap is installed
        void recordTap( const uint8_t* x8, const float* ts, int M, int K, int N ) const
        {
            ActivationTap* tap = activationTap();
            if ( !tap ) return;
            ActivationTap::Record r{ M, K, N, std::vector<uint8_t>( static_cast<size_t>( M ) * K ), std::vector<float>( static_cast<size_t>( M ) ) };
            rocmCheck( mila_cdna4_memcpy_d2h( r.x8.data(), x8, r.x8.size(), this->context_->getStream() ) );
            rocmCheck( mila_cdna4_memcpy_d2h( r.ts.data(), ts, r.ts.size() * 4, this->context_->getStream() ) );
            this->context_->synchronize();
            tap->records.push_back( std::move( r ) );
        }
        /// scratch for the per-token e4m3 activations + their scales (+ `extra` bytes behind them, 16-byte aligned: the GEMM's workspace) -- fetched per forward, never cached
        void activationScratch( int M, int K, uint8_t*& x8, float*& ts, size_t extra = 0, void** extra_out = nullptr ) const
        {
            const size_t xb = ( (size_t)M * K + 15 ) & ~(size_t)15, tb = ( (size_t)M * 4 + 15 ) & ~(size_t)15;
            auto* base = static_cast<uint8_t*>( this->context_->getScratch( xb + tb + extra ) );
            x8 = base; ts = reinterpret_cast<float*>( base + xb );
            if ( extra_out ) *extra_out = extra ? base + xb + tb : nullptr;
        }
        /// fp4 policy: W4A8 prefill (default on, as in the reference) or the dequantize -> bf16 GEMM fallback.
        /// fp8 policy: W8A8 prefill (default OFF: the reference's arithmetic for PerChannelFp8<> is W8A16, CudaLinearOp.ixx:597-644) -- the policy's e4m3 weights and
        /// per-channel scales consumed by the fp8 matrix cores where they lie (Policies.ixx:39-40); while it is on the op holds no bf16 copy of its weights
        void setFp8ActivationPrefill( bool on )
        {
            use_fp8_activation_prefill_ = on;
            if constexpr ( kFmt == 1 )
            {
                if ( on ) resident_bf16_.reset();
                else refreshResident();
            }
        }
        bool fp8ActivationPrefill() const noexcept { return use_fp8_activation_prefill_; }
        const float* weightFp8Scale() const noexcept { return weight_fp8_scale_ ? weight_fp8_scale_->data() : nullptr; }

        /// bytes of the op-owned resident staging (0 when off, when W8A8 is on, or for unquantized weights)
        size_t residentBytes() const noexcept { return ( resident_bf16_ ? resident_bf16_->size() * 2 : 0 ) + ( resident_e4m3_ ? resident_e4m3_->size() : 0 ); }
        /// op-owned device state right now: the resident staging + the fp4 policy's per-tensor e4m3 scale (CudaLinearOp.ixx:1118-1129 owns the same scalar)
        size_t stateBytes() const noexcept { return residentBytes() + ( weight_fp8_scale_ ? sizeof( float ) : 0 ); }
        /// ... and once quantized weights are in place, under the current switches (getRequiredStateMemorySize of the reference's ops)
        size_t requiredStateBytes() const noexcept
        {
            const size_t N = static_cast<size_t>( cfg_.out_features ), K = static_cast<size_t>( cfg_.in_features );
            if constexpr ( kFmt == 1 ) return ( resident_ && !use_fp8_activation_prefill_ ) ? N * K * 2 : 0;
            else if constexpr ( kFmt == 2 ) return sizeof( float ) + ( ( resident_ && K % 32 == 0 ) ? N * K : 0 );
            else return 0;
        }
        const void* weightPtr() const noexcept { return weight_; }
        const float* scalesPtr() const noexcept { return scales_; }
        const LinearOpConfig& config() const noexcept { return cfg_; }

    private:
        LinearOpConfig cfg_;
        const void* weight_{ nullptr };
        const uint16_t* bias_{ nullptr };
        const float* scales_{ nullptr };
        bool built_{ false };
        bool use_fp8_activation_prefill_{ kFmt == 2 };
        bool resident_{ true };
        std::unique_ptr<Tensor<TensorDataType::FP32, RocmDeviceMemoryResource>> weight_fp8_scale_;
        std::unique_ptr<RocmBf16Tensor> resident_bf16_;      // (quantized policies exist for BF16 activations only)
        std::unique_ptr<Tensor<TensorDataType::FP8_E4M3, RocmDeviceMemoryResource>> resident_e4m3_;

        void refreshResident()
        {
            if constexpr ( kFmt == 0 ) return;
            if ( !resident_ || !weight_ || !scales_ ) return;
            const int K = narrowToKernelIndex( cfg_.in_features, "in_features" ), N = narrowToKernelIndex( cfg_.out_features, "out_features" );
            mila_stream_t st = this->context_->getStream();
            // the shadows are an optimisation: if the device cannot hold them, fall back to per-forward staging (same results)
            try
            {
                if constexpr ( kFmt == 1 )
                {
                    if ( use_fp8_activation_prefill_ ) return;      // W8A8 reads the policy's own e4m3 weights: nothing to stage
                    if ( !resident_bf16_ ) resident_bf16_ = std::make_unique<RocmBf16Tensor>( this->context_->getDeviceId(), shape_t{ cfg_.out_features, cfg_.in_features } );
                }
                else if constexpr ( kFmt == 2 )
                {
                    if ( !weight_fp8_scale_ || K % 32 != 0 ) return;
                    if ( !resident_e4m3_ ) resident_e4m3_ = std::make_unique<Tensor<TensorDataType::FP8_E4M3, RocmDeviceMemoryResource>>( this->context_->getDeviceId(), shape_t{ cfg_.out_features, cfg_.in_features } );
                }
            }
            catch ( const std::exception& )
            {
                resident_ = false; resident_bf16_.reset(); resident_e4m3_.reset();
                return;
            }
            if constexpr ( kFmt == 1 ) rocmCheck( mila_cdna4_dequantize_to_bf16( resident_bf16_->data(), weight_, scales_, 1, N, K, 0, st ) );
            else if constexpr ( kFmt == 2 )
                rocmCheck( mila_cdna4_upcast_fp4_to_fp8( resident_e4m3_->data(), static_cast<const uint8_t*>( weight_ ), scales_, weight_fp8_scale_->data(), N, K, kGroup, st ) );
        }
    };

    template<typename TPolicy>
    struct OperationTraits<OperationType::LinearOp, DeviceType::Rocm, TensorDataType::BF16, TPolicy>
    {
        using type = RocmLinearOp<TensorDataType::BF16, TPolicy>;
    };
    /// the FP32 validation row: unquantized weights only (OperationTraits.Cuda.ixx:50-54)
    template<>
    struct OperationTraits<OperationType::LinearOp, DeviceType::Rocm, TensorDataType::FP32, Quant::Weight::NoWeightQuant>
    {
        using type = RocmLinearOp<TensorDataType::FP32, Quant::Weight::NoWeightQuant>;
    };
    // PerGroupInt4 has no row, like the quantize path of the reference (CudaLinearOp.ixx:385-391)
    template<int G>
    struct OperationTraits<OperationType::LinearOp, DeviceType::Rocm, TensorDataType::BF16, Quant::Weight::PerGroupInt4<G>>;

    // ---------------------------------------------------------------------------------------
    // RMSNorm / LayerNorm / Softmax / GELU / GeGLU / Residual
    // ---------------------------------------------------------------------------------------
    struct NormOpConfig
    {
        dim_t dim{ 0 };
        float epsilon{ 1e-5f };
        bool has_bias{ true };
        float weight_offset{ 0.0f };   ///< RmsNormConfig unit_offset
    };

    class RocmRmsNormOp : public Operation<DeviceType::Rocm, TensorDataType::BF16>
    {
    public:
        using TensorType = RocmBf16Tensor;
        RocmRmsNormOp( IExecutionContext* ctx, const NormOpConfig& cfg ) : Operation( ctx ), cfg_( cfg )
        {
            if ( cfg.dim <= 0 ) throw std::invalid_argument( "RocmRmsNormOp: normalized dimension must be positive" );
        }
        void setParameters( ITensor* weight, ITensor* bias )
        {
            w_ = weight ? static_cast<const uint16_t*>( weight->rawData() ) : nullptr;
            b_ = bias ? static_cast<const uint16_t*>( bias->rawData() ) : nullptr;
        }
        /// fixes the normalized (trailing) axis and the maximum slice count, and allocates the per-slice rstd the forward pass
        /// writes -- op-owned, bf16 like the reference's (RmsNormOp.ixx:174-262: rstd_tensor_ of num_slices elements)
        void build( const BuildContext& ctx )
        {
            if ( ctx.inputShape().back() != cfg_.dim ) throw std::invalid_argument( "RocmRmsNormOp::build: trailing dimension does not match the normalized shape" );
            const dim_t slices = shapeSize( ctx.inputShape() ) / cfg_.dim;
            rstd_ = std::make_unique<TensorType>( context_->getDeviceId(), shape_t{ slices } );
            built_ = true;
        }
        /// launch geometry follows the runtime tensor (built once at the prefill shape, decode arrives with one row, :271-300)
        void forward( const TensorType& in, TensorType& out ) const
        {
            if ( !built_ ) throw std::runtime_error( "RocmRmsNormOp::forward: not built" );
            if ( in.shape().empty() || in.shape().back() != cfg_.dim ) throw std::runtime_error( "RocmRmsNormOp::forward: input shape is incompatible with the built normalization axis" );
            const int dim = narrowToKernelIndex( cfg_.dim, "dim" );
            const int outer = narrowToKernelIndex( static_cast<dim_t>( in.size() ) / cfg_.dim, "outer" );
            if ( static_cast<size_t>( outer ) > rstd_->size() ) throw std::runtime_error( "RocmRmsNormOp::forward: runtime slice count exceeds the built maximum" );
            rocmCheck( mila_cdna4_rmsnorm_bf16( static_cast<uint16_t*>( out.rawData() ), rstd_->data(), static_cast<const uint16_t*>( in.rawData() ), w_, b_,
                                                outer, 1, dim, cfg_.epsilon, cfg_.weight_offset, context_->getStream() ) );
        }
        const TensorType* rstd() const noexcept { return rstd_.get(); }
        size_t stateBytes() const noexcept { return rstd_ ? rstd_->sizeInBytes() : 0; }
        const uint16_t* weightPtr() const noexcept { return w_; }
        float epsilon() const noexcept { return cfg_.epsilon; }
    private:
        NormOpConfig cfg_;
        const uint16_t* w_{ nullptr };
        const uint16_t* b_{ nullptr };
        std::unique_ptr<TensorType> rstd_;
        bool built_{ false };
    };
    template<> struct OperationTraits<OperationType::RmsNormOp, DeviceType::Rocm, TensorDataType::BF16> { using type = RocmRmsNormOp; };

    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    class RocmLayerNormOp : public Operation<DeviceType::Rocm, TPrecision>
    {
    public:
        using TensorType = RocmTensor<TPrecision>;
        RocmLayerNormOp( IExecutionContext* ctx, const NormOpConfig& cfg ) : Operation<DeviceType::Rocm, TPrecision>( ctx ), cfg_( cfg ) {}
        void setParameters( ITensor* weight, ITensor* bias )
        {
            w_ = weight ? weight->rawData() : nullptr;
            b_ = bias ? bias->rawData() : nullptr;
        }
        void build( const BuildContext& ) { built_ = true; }
        void forward( const TensorType& in, TensorType& out ) const
        {
            if ( !built_ ) throw std::runtime_error( "RocmLayerNormOp::forward: operation must be built before forward()" );
            const int dim = narrowToKernelIndex( cfg_.dim, "dim" );
            const int outer = narrowToKernelIndex( static_cast<dim_t>( in.size() ) / cfg_.dim, "outer" );
            if constexpr ( TPrecision == TensorDataType::FP32 )
                rocmCheck( mila_cdna4_layernorm_fp32( static_cast<float*>( out.rawData() ), nullptr, nullptr, static_cast<const float*>( in.rawData() ), static_cast<const float*>( w_ ),
                                                      static_cast<const float*>( b_ ), outer, dim, cfg_.epsilon, this->context_->getStream() ) );
            else
                rocmCheck( mila_cdna4_layernorm_bf16( static_cast<uint16_t*>( out.rawData() ), nullptr, nullptr, static_cast<const uint16_t*>( in.rawData() ),
                                                      static_cast<const uint16_t*>( w_ ), static_cast<const uint16_t*>( b_ ), outer, dim, cfg_.epsilon, this->context_->getStream() ) );
        }
    private:
        NormOpConfig cfg_;
        const void* w_{ nullptr };
        const void* b_{ nullptr };
        bool built_{ false };
    };
    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    struct OperationTraits<OperationType::LayerNormOp, DeviceType::Rocm, TPrecision> { using type = RocmLayerNormOp<TPrecision>; };

    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    class RocmGeluOp : public Operation<DeviceType::Rocm, TPrecision>
    {
    public:
        using TensorType = RocmTensor<TPrecision>;
        explicit RocmGeluOp( IExecutionContext* ctx ) : Operation<DeviceType::Rocm, TPrecision>( ctx ) {}
        void forward( const TensorType& in, TensorType& out ) const
        {
            if constexpr ( TPrecision == TensorDataType::FP32 )
                rocmCheck( mila_cdna4_gelu_fp32( static_cast<float*>( out.rawData() ), static_cast<const float*>( in.rawData() ), static_cast<int64_t>( in.size() ), this->context_->getStream() ) );
            else
                rocmCheck( mila_cdna4_gelu_bf16( static_cast<uint16_t*>( out.rawData() ), static_cast<const uint16_t*>( in.rawData() ),
                                                 static_cast<int64_t>( in.size() ), this->context_->getStream() ) );
        }
    };
    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    struct OperationTraits<OperationType::GeluOp, DeviceType::Rocm, TPrecision> { using type = RocmGeluOp<TPrecision>; };

    /// Swiglu<..., Gelu> resolves to the GeGLU op (Gemma.Block.ixx:150)
    class RocmGegluOp : public Operation<DeviceType::Rocm, TensorDataType::BF16>
    {
    public:
        using TensorType = RocmBf16Tensor;
        explicit RocmGegluOp( IExecutionContext* ctx ) : Operation( ctx ) {}
        void forward( const TensorType& in, TensorType& out ) const
        {
            const dim_t two_h = in.shape().back();
            if ( two_h % 2 != 0 ) throw std::invalid_argument( "RocmGegluOp::forward: the last dimension must be even ([gate | up])" );
            const int half = narrowToKernelIndex( two_h / 2, "half" );
            const int tokens = narrowToKernelIndex( static_cast<dim_t>( in.size() ) / two_h, "tokens" );
            rocmCheck( mila_cdna4_geglu_bf16( static_cast<uint16_t*>( out.rawData() ), static_cast<const uint16_t*>( in.rawData() ), tokens, half, context_->getStream() ) );
        }
    };
    template<> struct OperationTraits<OperationType::GegluOp, DeviceType::Rocm, TensorDataType::BF16> { using type = RocmGegluOp; };

    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    class RocmResidualOp : public Operation<DeviceType::Rocm, TPrecision>
    {
    public:
        using TensorType = RocmTensor<TPrecision>;
        explicit RocmResidualOp( IExecutionContext* ctx ) : Operation<DeviceType::Rocm, TPrecision>( ctx ) {}
        void forward( const TensorType& a, const TensorType& b, TensorType& out ) const
        {
            if ( a.size() != b.size() ) throw std::invalid_argument( "RocmResidualOp::forward: operand sizes differ" );
            if constexpr ( TPrecision == TensorDataType::FP32 )
                rocmCheck( mila_cdna4_residual_fp32( static_cast<float*>( out.rawData() ), static_cast<const float*>( a.rawData() ), static_cast<const float*>( b.rawData() ),
                                                     static_cast<int64_t>( a.size() ), this->context_->getStream() ) );
            else
                rocmCheck( mila_cdna4_residual_bf16( static_cast<uint16_t*>( out.rawData() ), static_cast<const uint16_t*>( a.rawData() ),
                                                     static_cast<const uint16_t*>( b.rawData() ), static_cast<int64_t>( a.size() ), this->context_->getStream() ) );
        }
    };
    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    struct OperationTraits<OperationType::ResidualOp, DeviceType::Rocm, TPrecision> { using type = RocmResidualOp<TPrecision>; };

    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    class RocmSoftmaxOp : public Operation<DeviceType::Rocm, TPrecision>
    {
    public:
        using TensorType = RocmTensor<TPrecision>;
        RocmSoftmaxOp( IExecutionContext* ctx, int axis ) : Operation<DeviceType::Rocm, TPrecision>( ctx ), axis_( axis ) {}
        void forward( const TensorType& in, TensorType& out ) const
        {
            const auto& s = in.shape();
            const int rank = static_cast<int>( s.size() );
            const int ax = axis_ < 0 ? axis_ + rank : axis_;
            if ( ax < 0 || ax >= rank ) throw std::invalid_argument( "RocmSoftmaxOp::forward: axis out of range" );
            dim_t outer = 1, inner = 1;
            for ( int i = 0; i < ax; ++i ) outer *= s[ i ];
            for ( int i = ax + 1; i < rank; ++i ) inner *= s[ i ];
            if constexpr ( TPrecision == TensorDataType::FP32 )
                rocmCheck( mila_cdna4_softmax_fp32( static_cast<float*>( out.rawData() ), static_cast<const float*>( in.rawData() ), narrowToKernelIndex( outer, "outer" ),
                                                    narrowToKernelIndex( s[ ax ], "dim" ), narrowToKernelIndex( inner, "inner" ), this->context_->getStream() ) );
            else
                rocmCheck( mila_cdna4_softmax_bf16( static_cast<uint16_t*>( out.rawData() ), static_cast<const uint16_t*>( in.rawData() ),
                                                    narrowToKernelIndex( outer, "outer" ), narrowToKernelIndex( s[ ax ], "dim" ),
                                                    narrowToKernelIndex( inner, "inner" ), this->context_->getStream() ) );
        }
    private:
        int axis_;
    };
    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    struct OperationTraits<OperationType::SoftmaxOp, DeviceType::Rocm, TPrecision> { using type = RocmSoftmaxOp<TPrecision>; };

    // ---------------------------------------------------------------------------------------
    // RoPE (IPositionalPairedOp) -- cache built once per (max_seq, head_dim, base, rotary_dim)
    // ---------------------------------------------------------------------------------------
    struct RopeOpConfig
    {
        dim_t max_seq{ 0 }, head_dim{ 0 }, num_heads{ 0 }, num_kv_heads{ 0 };
        float base{ 10000.0f };
        dim_t rotary_dim{ 0 };
    };

    /// Process-wide cos / sin tables, one pair per distinct (device, max_seq, head_dim, base, rotary_dim): every layer of a model that rotates with the same geometry
    /// shares ONE pair (the reference's RopeCacheRegistry, Gemma.ixx:455-462: "only one per distinct key is ever allocated: Gemma has two, the local and global theta").
    /// Entries are weak: the tables live as long as an op holds them.
    class RopeCacheRegistry
    {
    public:
        using CacheTensor = Tensor<TensorDataType::FP32, RocmDeviceMemoryResource>;
        struct Tables { std::shared_ptr<CacheTensor> cos, sin; };
        /// the shared pair for this key; `created` says whether the caller must fill it
        static Tables acquire( DeviceId dev, const std::tuple<int, dim_t, dim_t, uint32_t, dim_t>& key, dim_t max_seq, dim_t head_dim, bool& created )
        {
            static std::mutex mu;
            static std::map<std::tuple<int, dim_t, dim_t, uint32_t, dim_t>, std::pair<std::weak_ptr<CacheTensor>, std::weak_ptr<CacheTensor>>> entries;
            std::lock_guard<std::mutex> lock( mu );
            auto& e = entries[ key ];
            Tables t{ e.first.lock(), e.second.lock() };
            created = !t.cos || !t.sin;
            if ( created )
            {
                t.cos = std::make_shared<CacheTensor>( dev, shape_t{ max_seq, head_dim / 2 } );
                t.sin = std::make_shared<CacheTensor>( dev, shape_t{ max_seq, head_dim / 2 } );
                e = { t.cos, t.sin };
            }
            return t;
        }
    };

    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    class RocmRopeOp : public Operation<DeviceType::Rocm, TPrecision>
    {
        using Operation<DeviceType::Rocm, TPrecision>::context_;
    public:
        using TensorType = RocmTensor<TPrecision>;
        using CacheTensor = Tensor<TensorDataType::FP32, RocmDeviceMemoryResource>;
        RocmRopeOp( IExecutionContext* ctx, const RopeOpConfig& cfg ) : Operation<DeviceType::Rocm, TPrecision>( ctx ), cfg_( cfg )
        {
            if ( cfg.head_dim <= 0 || cfg.head_dim % 2 != 0 ) throw std::invalid_argument( "RocmRopeOp: head_dim must be positive and even" );
            if ( cfg.max_seq <= 0 ) throw std::invalid_argument( "RocmRopeOp: max_seq must be positive" );
        }
        void build( const BuildContext& )
        {
            uint32_t base_bits;
            std::memcpy( &base_bits, &cfg_.base, 4 );
            bool created = false;
            auto t = RopeCacheRegistry::acquire( context_->getDeviceId(), { context_->getDeviceId().index, cfg_.max_seq, cfg_.head_dim, base_bits, cfg_.rotary_dim }, cfg_.max_seq, cfg_.head_dim, created );
            cos_ = t.cos; sin_ = t.sin;
            if ( created )
            {
                rocmCheck( mila_cdna4_rope_build_cache( cos_->data(), sin_->data(), narrowToKernelIndex( cfg_.max_seq, "max_seq" ),
                                                        narrowToKernelIndex( cfg_.head_dim, "head_dim" ), cfg_.base,
                                                        narrowToKernelIndex( cfg_.rotary_dim, "rotary_dim" ), context_->getStream() ) );
                context_->synchronize();      // another op (another context's stream) may read the shared tables next
            }
            built_ = true;
        }
        /// what ONE owner of these tables pays (Rope.ixx:254: the operation reports one owner's bytes; a composite whose layers share a key subtracts the duplicates)
        size_t stateBytes() const noexcept { return ( cos_ ? cos_->sizeInBytes() : 0 ) + ( sin_ ? sin_->sizeInBytes() : 0 ); }
        size_t requiredStateBytes() const noexcept { return 2 * static_cast<size_t>( cfg_.max_seq ) * static_cast<size_t>( cfg_.head_dim / 2 ) * sizeof( float ); }
        const void* tableKey() const noexcept { return cos_.get(); }      ///< equal for ops that share one pair
        /// rotate q [B,T,NH,HS] and k [B,T,NKV,HS] in place (Components/Encodings/Rope/Rope.ixx:107,160-200)
        void prefill( TensorType& q, TensorType& k, int B, int T, int position_offset ) const
        {
            if ( !built_ ) throw std::runtime_error( "RocmRopeOp: not built" );
            if constexpr ( TPrecision == TensorDataType::FP32 )
            {
                auto* qp = static_cast<float*>( q.rawData() );
                auto* kp = static_cast<float*>( k.rawData() );
                rocmCheck( mila_cdna4_rope_forward_fp32( qp, kp, qp, kp, cos_->data(), sin_->data(), B, T, narrowToKernelIndex( cfg_.num_heads, "NH" ),
                                                         narrowToKernelIndex( cfg_.num_kv_heads, "NKV" ), narrowToKernelIndex( cfg_.head_dim, "HS" ),
                                                         position_offset, narrowToKernelIndex( cfg_.max_seq, "max_seq" ), context_->getStream() ) );
            }
            else
            {
                auto* qp = static_cast<uint16_t*>( q.rawData() );
                auto* kp = static_cast<uint16_t*>( k.rawData() );
                rocmCheck( mila_cdna4_rope_forward_bf16( qp, kp, qp, kp, cos_->data(), sin_->data(), B, T, narrowToKernelIndex( cfg_.num_heads, "NH" ),
                                                         narrowToKernelIndex( cfg_.num_kv_heads, "NKV" ), narrowToKernelIndex( cfg_.head_dim, "HS" ),
                                                         position_offset, narrowToKernelIndex( cfg_.max_seq, "max_seq" ), context_->getStream() ) );
            }
        }
        void decode( TensorType& q, TensorType& k, int B, int position ) const { prefill( q, k, B, 1, position ); }
        const float* cosCache() const noexcept { return cos_ ? cos_->data() : nullptr; }
        const float* sinCache() const noexcept { return sin_ ? sin_->data() : nullptr; }
    private:
        RopeOpConfig cfg_;
        std::shared_ptr<CacheTensor> cos_, sin_;
        bool built_{ false };
    };
    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    struct OperationTraits<OperationType::RopeOp, DeviceType::Rocm, TPrecision> { using type = RocmRopeOp<TPrecision>; };

    // ---------------------------------------------------------------------------------------
    // Grouped-query attention over an op-owned KV cache (IKvInference)
    // ---------------------------------------------------------------------------------------
    struct GqaOpConfig
    {
        dim_t num_heads{ 0 }, num_kv_heads{ 0 }, head_dim{ 0 };
        dim_t window{ 0 };               ///< 0 = global
        float attention_scale{ 0.0f };   ///< <= 0 -> 1/sqrt(head_dim)  (GroupedQueryAttention.Config.ixx:190-200)
    };

    /// Shared body of the two KV-cache policies of CudaGqaOp<Prec, kBounded> (OPS/Attention/GQA/CudaGqaOp.ixx:97-985): the
    /// policy only changes the cache capacity rule and what rewind may restore, so a model can hold bounded (sliding-window)
    /// and unbounded (global) layers behind one pointer type.
    class RocmGqaOpBase : public Operation<DeviceType::Rocm, TensorDataType::BF16>
    {
    public:
        using TensorType = RocmBf16Tensor;
        RocmGqaOpBase( IExecutionContext* ctx, const GqaOpConfig& cfg, bool bounded_ring ) : Operation( ctx ), cfg_( cfg ), kBoundedRing( bounded_ring )
        {
            if ( cfg.num_heads <= 0 || cfg.num_kv_heads <= 0 || cfg.num_heads % cfg.num_kv_heads != 0 )
                throw std::invalid_argument( "RocmGqaOp: num_heads must be a positive multiple of num_kv_heads" );
            if ( cfg.head_dim <= 0 ) throw std::invalid_argument( "RocmGqaOp: head_dim must be positive" );
            if ( kBoundedRing && cfg.window <= 0 ) throw std::invalid_argument( "RocmGqaOp: a bounded ring cache needs a sliding window" );
        }
        virtual ~RocmGqaOpBase() = default;
        bool boundedRing() const noexcept { return kBoundedRing; }

        float scale() const noexcept { return cfg_.attention_scale > 0.0f ? cfg_.attention_scale : 1.0f / std::sqrt( static_cast<float>( cfg_.head_dim ) ); }

        /// capacity rule of CudaGqaOp::resolveCacheCapacity (CudaGqaOp.ixx:552-574)
        dim_t resolveCacheCapacity( dim_t max_seq, dim_t window, dim_t prefill_chunk ) const
        {
            if ( kBoundedRing ) return std::min( max_seq, window + std::max<dim_t>( prefill_chunk, 1 ) - 1 );
            return max_seq;
        }

        void initializeKvCache( int batch, dim_t max_seq, dim_t prefill_chunk )
        {
            batch_ = batch;
            capacity_ = resolveCacheCapacity( max_seq, cfg_.window, prefill_chunk );
            const shape_t s{ batch, cfg_.num_kv_heads, capacity_, cfg_.head_dim };
            k_cache_ = std::make_unique<TensorType>( context_->getDeviceId(), s );
            v_cache_ = std::make_unique<TensorType>( context_->getDeviceId(), s );
            length_ = 0;
        }
        void resetKvCache() noexcept { length_ = 0; }
        void rewindKvCache( dim_t length )
        {
            if ( length < 0 || length > length_ ) throw std::invalid_argument( "RocmGqaOp::rewindKvCache: bad length" );
            // ring validity as the reference states it (CudaGqaOp.ixx:182-203): a continuation from `length` attends down to length - window,
            // so the stale tail [length, length_) must not have wrapped over that range: at most capacity - window (= chunk - 1) stale tokens
            if ( kBoundedRing && length_ - length > capacity_ - cfg_.window ) throw std::runtime_error( "RocmGqaOp::rewindKvCache: evicted positions cannot be restored" );
            length_ = length;
        }
        dim_t cacheLength() const noexcept { return length_; }
        /// a caller that appended through the fused entry points (which take the cache pointers directly) reports how far it wrote
        void noteCacheLength( dim_t length ) { if ( length < 0 ) throw std::invalid_argument( "RocmGqaOp::noteCacheLength: negative length" ); length_ = length; }
        dim_t cacheCapacity() const noexcept { return capacity_; }

        /// q [B,chunk,NH*HS], k/v [B,chunk,NKV*HS] at absolute positions [position, position+chunk)
        void prefill( const TensorType& q, const TensorType& k, const TensorType& v, TensorType& out, int chunk, int position )
        {
            requireCache();
            mila_stream_t st = context_->getStream();
            const int NH = (int)cfg_.num_heads, NKV = (int)cfg_.num_kv_heads, HS = (int)cfg_.head_dim, cap = (int)capacity_;
            rocmCheck( mila_cdna4_kv_write_bf16( k_cache_->data(), v_cache_->data(), q_cast( k ), q_cast( v ), batch_, chunk, NKV, HS, position, cap, st ) );
            rocmCheck( mila_cdna4_attn_prefill_bf16( out.data(), q_cast( q ), k_cache_->data(), v_cache_->data(), batch_, chunk, NH, NKV, HS, cap,
                                                     position, (int)cfg_.window, scale(), st ) );
            length_ = position + chunk;
        }

        /// the chunk's K/V rows are already in the cache (written by the fused q/k/v post-processing kernel):
        /// attention only.  q [B,chunk,NH*HS] at absolute positions [position, position+chunk)
        void prefillFromCache( const TensorType& q, TensorType& out, int chunk, int position )
        {
            requireCache();
            const int NH = (int)cfg_.num_heads, NKV = (int)cfg_.num_kv_heads, HS = (int)cfg_.head_dim, cap = (int)capacity_;
            rocmCheck( mila_cdna4_attn_prefill_bf16( out.data(), q_cast( q ), k_cache_->data(), v_cache_->data(), batch_, chunk, NH, NKV, HS, cap,
                                                     position, (int)cfg_.window, scale(), context_->getStream() ) );
            length_ = position + chunk;
        }

        /// one token per sequence at absolute position `position`
        void decode( const TensorType& q, const TensorType& k, const TensorType& v, TensorType& out, int position )
        {
            requireCache();
            mila_stream_t st = context_->getStream();
            const int NKV = (int)cfg_.num_kv_heads, HS = (int)cfg_.head_dim, cap = (int)capacity_;
            rocmCheck( mila_cdna4_kv_write_bf16( k_cache_->data(), v_cache_->data(), q_cast( k ), q_cast( v ), batch_, 1, NKV, HS, position, cap, st ) );
            attendDecode( q, out, position );
        }

        /// attention only (the fused q/k/v post-processing kernel has already appended K/V)
        void attendDecode( const TensorType& q, TensorType& out, int position )
        {
            requireCache();
            const int NH = (int)cfg_.num_heads, NKV = (int)cfg_.num_kv_heads, HS = (int)cfg_.head_dim, cap = (int)capacity_;
            const size_t need = mila_cdna4_attn_decode_scratch_bytes( batch_, NH, HS );
            void* scratch = context_->getScratch( need );   // fetched per forward, never cached
            rocmCheck( mila_cdna4_attn_decode_bf16( out.data(), q_cast( q ), k_cache_->data(), v_cache_->data(), scratch, need, batch_, NH, NKV, HS, cap,
                                                    position + 1, (int)cfg_.window, scale(), context_->getStream() ) );
            length_ = position + 1;
        }

        size_t stateBytes() const noexcept { return ( k_cache_ ? k_cache_->sizeInBytes() : 0 ) + ( v_cache_ ? v_cache_->sizeInBytes() : 0 ); }
        size_t requiredStateBytes( int batch, dim_t max_seq, dim_t prefill_chunk ) const
        {
            return 2 * static_cast<size_t>( batch ) * static_cast<size_t>( cfg_.num_kv_heads ) * static_cast<size_t>( resolveCacheCapacity( max_seq, cfg_.window, prefill_chunk ) ) *
                   static_cast<size_t>( cfg_.head_dim ) * 2;
        }
        uint16_t* keyCache() noexcept { return k_cache_ ? k_cache_->data() : nullptr; }
        uint16_t* valueCache() noexcept { return v_cache_ ? v_cache_->data() : nullptr; }
        const GqaOpConfig& config() const noexcept { return cfg_; }

    private:
        static const uint16_t* q_cast( const TensorType& t ) { return static_cast<const uint16_t*>( t.rawData() ); }
        void requireCache() const { if ( !k_cache_ ) throw std::runtime_error( "RocmGqaOp: initializeKvCache() must be called first" ); }
        GqaOpConfig cfg_;
        const bool kBoundedRing;
        std::unique_ptr<TensorType> k_cache_, v_cache_;
        int batch_{ 1 };
        dim_t capacity_{ 0 }, length_{ 0 };
    };
    /// counterpart of CudaGqaOp<Prec, kBounded>: the KV policy as a compile-time axis, as in the reference
    template<bool kBoundedRingPolicy>
    class RocmGqaOp : public RocmGqaOpBase
    {
    public:
        RocmGqaOp( IExecutionContext* ctx, const GqaOpConfig& cfg ) : RocmGqaOpBase( ctx, cfg, kBoundedRingPolicy ) {}
    };
    template<typename TKvPolicy>
    struct OperationTraits<OperationType::GroupedQueryAttentionOp, DeviceType::Rocm, TensorDataType::BF16, TKvPolicy>
    {
        using type = RocmGqaOp<TKvPolicy::kBoundedRing>;
    };

    // ---------------------------------------------------------------------------------------
    // Sampling (row f3): counterpart of CudaSamplingOp<FP32> (OPS/Sampling/CudaSamplingOp.ixx; Tests/Dnn/Samplers/Sampling.Cuda.cpp:40-150).
    // forward(): sample on the context's stream into a device token.  enqueueForward() / awaitToken(): the decode-ahead half of the pipelined generation loop --
    // the token is ALSO published to a host-visible slot by the sampler's stream (no host synchronize between forward and sample), awaitToken() blocks until it is
    // there; awaitToken() without an outstanding enqueueForward() is a caller bug: std::logic_error, not UB (Sampling.Cuda.cpp:503-509).
    // ---------------------------------------------------------------------------------------
    struct SamplingOpConfig { dim_t vocab_size{ 0 }; float final_logit_softcap{ 0.0f }; };
    struct SamplingParams { float temperature{ 1.0f }; int top_k{ 0 }; float top_p{ 1.0f }; };      ///< Components/Transformers/SamplingParams.ixx: temperature <= 0 = greedy

    class RocmSamplingOp : public Operation<DeviceType::Rocm, TensorDataType::FP32>
    {
    public:
        using LogitsTensor = Tensor<TensorDataType::FP32, RocmDeviceMemoryResource>;
        using TokenTensor = Tensor<TensorDataType::INT32, RocmDeviceMemoryResource>;
        RocmSamplingOp( IExecutionContext* ctx, const SamplingOpConfig& cfg ) : Operation( ctx ), cfg_( cfg )
        {
            if ( cfg.vocab_size <= 0 ) throw std::invalid_argument( "RocmSamplingOp: vocabulary size must be positive" );
            scratch_bytes_ = std::max( mila_cdna4_sample_scratch_bytes(), mila_cdna4_sample_stochastic_scratch_bytes( narrowToKernelIndex( cfg.vocab_size, "vocab" ) ) );
            scratch_ = std::make_unique<Tensor<TensorDataType::UINT8, RocmDeviceMemoryResource>>( context_->getDeviceId(), shape_t{ static_cast<dim_t>( scratch_bytes_ ) } );
            seq_dev_ = std::make_unique<Tensor<TensorDataType::FP32, RocmDeviceMemoryResource>>( context_->getDeviceId(), shape_t{ 2 } );      // one 64-bit counter
            rocmCheck( mila_cdna4_memset_zero( seq_dev_->rawData(), 8, context_->getStream() ) );
            void* host = nullptr;
            rocmCheck( mila_cdna4_host_alloc_pinned( &host, sizeof( unsigned long long ) * kSlots ) );
            ring_ = static_cast<unsigned long long*>( host );
            for ( int i = 0; i < kSlots; ++i ) ring_[ i ] = 0;
            context_->synchronize();
        }
        ~RocmSamplingOp() override { if ( ring_ ) mila_cdna4_host_free_pinned( ring_ ); }

        /// logits [.., vocab] -> token_out[0]; r in [0, 1) is drawn by the caller (the reference injects it the same way)
        void forward( const LogitsTensor& logits, TokenTensor& token_out, const SamplingParams& sp, float r ) const
        {
            const int V = narrowToKernelIndex( cfg_.vocab_size, "vocab" );
            if ( static_cast<dim_t>( logits.size() ) < cfg_.vocab_size ) throw std::invalid_argument( "RocmSamplingOp::forward: logits are shorter than the vocabulary" );
            if ( sp.temperature <= 0.0f || sp.top_k == 1 )
                rocmCheck( mila_cdna4_sample_argmax_fp32( logits.data(), token_out.data(), V, scratch_->rawData(), scratch_bytes_, context_->getStream() ) );
            else
                rocmCheck( mila_cdna4_sample_stochastic_fp32( logits.data(), token_out.data(), V, cfg_.final_logit_softcap, sp.temperature, sp.top_k, sp.top_p, r, scratch_->rawData(),
                                                              scratch_bytes_, context_->getStream() ) );
        }
        void enqueueForward( const LogitsTensor& logits, TokenTensor& token_out, const SamplingParams& sp, float r )
        {
            forward( logits, token_out, sp, r );
            rocmCheck( mila_cdna4_snapshot_token( token_out.data(), reinterpret_cast<unsigned long long*>( seq_dev_->rawData() ), ring_, kSlots, context_->getStream() ) );
            ++enqueued_;
        }
        int32_t awaitToken()
        {
            if ( awaited_ >= enqueued_ ) throw std::logic_error( "RocmSamplingOp::awaitToken: no outstanding enqueueForward()" );
            const uint64_t seq = ++awaited_;
            volatile unsigned long long* slot = ring_ + ( seq % kSlots );
            const auto t0 = std::chrono::steady_clock::now();
            for ( uint64_t spins = 0;; ++spins )
            {
                const unsigned long long v = __atomic_load_n( slot, __ATOMIC_ACQUIRE );
                if ( ( v >> 32 ) == ( seq & 0xffffffffull ) ) return static_cast<int32_t>( static_cast<uint32_t>( v ) );
                if ( ( spins & 1023 ) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds( 20 ) )
                    throw std::runtime_error( "RocmSamplingOp::awaitToken: the device did not publish a token within 20 s" );
            }
        }
    private:
        static constexpr int kSlots = 4;
        SamplingOpConfig cfg_;
        size_t scratch_bytes_{ 0 };
        std::unique_ptr<Tensor<TensorDataType::UINT8, RocmDeviceMemoryResource>> scratch_;
        std::unique_ptr<Tensor<TensorDataType::FP32, RocmDeviceMemoryResource>> seq_dev_;
        unsigned long long* ring_{ nullptr };
        uint64_t enqueued_{ 0 }, awaited_{ 0 };
    };
    template<> struct OperationTraits<OperationType::SamplingOp, DeviceType::Rocm, TensorDataType::FP32> { using type = RocmSamplingOp; };

    /// GPT-2 attention on packed QKV (new BF16 row; the reference's CUDA MHA is FP32-only, OPS/OperationTraits.Cuda.ixx:274-282) with the KV-cache
    /// interface of CudaMultiHeadAttentionOp (OPS/Attention/MHA/CudaMhaOp.ixx:107-380: IPositionalUnaryOp::prefill / decode + IKvCacheLifecycle)
    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    class RocmMultiHeadAttentionOp : public Operation<DeviceType::Rocm, TPrecision>
    {
        using Operation<DeviceType::Rocm, TPrecision>::context_;
        static constexpr bool kFp32 = TPrecision == TensorDataType::FP32;      // the reference's own (and only) CUDA MHA row is FP32 (OperationTraits.Cuda.ixx:274-282)
    public:
        using TensorType = RocmTensor<TPrecision>;
        RocmMultiHeadAttentionOp( IExecutionContext* ctx, dim_t model_dim, dim_t num_heads ) : Operation<DeviceType::Rocm, TPrecision>( ctx ), C_( model_dim ), NH_( num_heads )
        {
            if ( model_dim <= 0 || num_heads <= 0 || model_dim % num_heads != 0 ) throw std::invalid_argument( "RocmMultiHeadAttentionOp: model_dim must be a positive multiple of num_heads" );
        }
        /// the built [B, T, 3C] shape bounds every later call (CudaMhaOp.ixx:423-447)
        void build( const BuildContext& ctx )
        {
            const auto& s = ctx.inputShape();
            if ( s.size() != 3 || s[ 2 ] != 3 * C_ ) throw std::invalid_argument( "RocmMultiHeadAttentionOp::build: expected [B, T, 3C]" );
            B_ = s[ 0 ]; T_ = s[ 1 ];
            cached_seq_len_ = 0; kv_cache_enabled_ = false;
        }
        void forward( const TensorType& qkv, TensorType& out ) const
        {
            const auto& s = qkv.shape();
            if ( s.size() != 3 || s[ 2 ] != 3 * C_ ) throw std::invalid_argument( "RocmMultiHeadAttentionOp::forward: expected [B, T, 3C]" );
            if constexpr ( kFp32 ) rocmCheck( mila_cdna4_mha_fp32( static_cast<float*>( out.rawData() ), static_cast<const float*>( qkv.rawData() ), (int)s[ 0 ], (int)s[ 1 ], (int)C_, (int)NH_, context_->getStream() ) );
            else rocmCheck( mila_cdna4_mha_bf16( static_cast<uint16_t*>( out.rawData() ), static_cast<const uint16_t*>( qkv.rawData() ), (int)s[ 0 ], (int)s[ 1 ], (int)C_, (int)NH_, context_->getStream() ) );
        }

        // ---- IKvCacheLifecycle (CudaMhaOp.ixx:112-143) ----
        void initializeKvCache( dim_t batch_size, dim_t max_sequence_length )
        {
            if ( batch_size != B_ ) throw std::invalid_argument( "RocmMultiHeadAttentionOp::initializeKvCache batch size must match the built shape" );
            if ( max_sequence_length <= 0 || max_sequence_length > T_ ) throw std::invalid_argument( "RocmMultiHeadAttentionOp::initializeKvCache max_sequence_length out of range" );
            if ( !k_cache_ || active_max_seq_len_ != max_sequence_length )
            {
                const shape_t cs{ B_, NH_, max_sequence_length, C_ / NH_ };
                k_cache_ = std::make_unique<TensorType>( context_->getDeviceId(), cs );
                v_cache_ = std::make_unique<TensorType>( context_->getDeviceId(), cs );
            }
            active_max_seq_len_ = max_sequence_length;
            cached_seq_len_ = 0;
            kv_cache_enabled_ = true;
        }
        void resetKvCache() noexcept { cached_seq_len_ = 0; }
        bool rewindKvCache( dim_t position ) noexcept
        {
            if ( position < 0 || position > cached_seq_len_ ) return false;
            cached_seq_len_ = position;
            return true;
        }
        dim_t cacheLength() const noexcept { return cached_seq_len_; }
        size_t stateBytes() const noexcept { return ( k_cache_ ? k_cache_->sizeInBytes() : 0 ) + ( v_cache_ ? v_cache_->sizeInBytes() : 0 ); }

        // ---- IPositionalUnaryOp ----
        /// the whole prompt [B, T' <= max_seq, 3C]: causal attention + the prompt's K / V rows into the cache (CudaMhaOp.ixx:145-232)
        void prefill( const TensorType& qkv, TensorType& out )
        {
            ensureKvCacheEnabled();
            const auto& s = qkv.shape();
            if ( s.size() != 3 || s[ 0 ] != B_ || s[ 2 ] != 3 * C_ || s[ 1 ] <= 0 || s[ 1 ] > active_max_seq_len_ )
                throw std::invalid_argument( "RocmMultiHeadAttentionOp::prefill: input must be [B, T <= max_sequence_length, 3C]" );
            mila_stream_t st = context_->getStream();
            if constexpr ( kFp32 )
            {
                const auto* x = static_cast<const float*>( qkv.rawData() );
                rocmCheck( mila_cdna4_mha_fp32( static_cast<float*>( out.rawData() ), x, (int)B_, (int)s[ 1 ], (int)C_, (int)NH_, st ) );
                rocmCheck( mila_cdna4_mha_kv_write_fp32( static_cast<float*>( k_cache_->rawData() ), static_cast<float*>( v_cache_->rawData() ), x, (int)B_, (int)s[ 1 ], (int)C_, (int)NH_, 0,
                                                         (int)active_max_seq_len_, st ) );
            }
            else
            {
                const auto* x = static_cast<const uint16_t*>( qkv.rawData() );
                rocmCheck( mila_cdna4_mha_bf16( static_cast<uint16_t*>( out.rawData() ), x, (int)B_, (int)s[ 1 ], (int)C_, (int)NH_, st ) );
                rocmCheck( mila_cdna4_mha_kv_write_bf16( static_cast<uint16_t*>( k_cache_->rawData() ), static_cast<uint16_t*>( v_cache_->rawData() ), x, (int)B_, (int)s[ 1 ], (int)C_, (int)NH_, 0,
                                                         (int)active_max_seq_len_, st ) );
            }
            cached_seq_len_ = s[ 1 ];
        }
        /// one token per sequence [B, 1, 3C] at absolute `position`: appends its K / V, attends to keys 0 .. position (CudaMhaOp.ixx:252-380)
        void decode( const TensorType& qkv, TensorType& out, dim_t position )
        {
            ensureKvCacheEnabled();
            const auto& s = qkv.shape();
            if ( s.size() != 3 || s[ 0 ] != B_ || s[ 1 ] != 1 || s[ 2 ] != 3 * C_ ) throw std::invalid_argument( "RocmMultiHeadAttentionOp::decode: input must be [B, 1, 3C]" );
            if ( position < 0 || position >= active_max_seq_len_ ) throw std::invalid_argument( "RocmMultiHeadAttentionOp::decode position out of range" );
            if constexpr ( kFp32 )
                rocmCheck( mila_cdna4_mha_decode_fp32( static_cast<float*>( out.rawData() ), static_cast<const float*>( qkv.rawData() ), static_cast<float*>( k_cache_->rawData() ),
                                                       static_cast<float*>( v_cache_->rawData() ), (int)B_, (int)C_, (int)NH_, (int)active_max_seq_len_, (int)position, context_->getStream() ) );
            else
            {
                const size_t need = mila_cdna4_mha_decode_scratch_bytes( (int)B_, (int)C_, (int)NH_ );
                void* scratch = context_->getScratch( need );      // fetched per call, never cached
                rocmCheck( mila_cdna4_mha_decode_bf16( static_cast<uint16_t*>( out.rawData() ), static_cast<const uint16_t*>( qkv.rawData() ), static_cast<uint16_t*>( k_cache_->rawData() ),
                                                       static_cast<uint16_t*>( v_cache_->rawData() ), scratch, need, (int)B_, (int)C_, (int)NH_, (int)active_max_seq_len_, (int)position,
                                                       context_->getStream() ) );
            }
            if ( position + 1 > cached_seq_len_ ) cached_seq_len_ = position + 1;
        }
    private:
        void ensureKvCacheEnabled() const { if ( !kv_cache_enabled_ ) throw std::runtime_error( "RocmMultiHeadAttentionOp: initializeKvCache() must be called first" ); }
        dim_t C_, NH_;
        dim_t B_{ 0 }, T_{ 0 }, active_max_seq_len_{ 0 }, cached_seq_len_{ 0 };
        bool kv_cache_enabled_{ false };
        std::unique_ptr<TensorType> k_cache_, v_cache_;
    };
    template<TensorDataType TPrecision> requires RocmPrecision<TPrecision>
    struct OperationTraits<OperationType::MultiHeadAttentionOp, DeviceType::Rocm, TPrecision> { using type = RocmMultiHeadAttentionOp<TPrecision>; };
}
