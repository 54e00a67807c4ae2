// C entry points for ONE Linear<Rocm, BF16, TWeightQuant> component of the host mirror (tests: the dispatch-ladder sweep drives RocmLinearOp::forward at every row
// count, tests/test_dispatch_ladder_gpu.py; the reference's pre-quantized-artifact scenarios, Tests/Dnn/Components/Linear/Linear.Cuda.cpp:1145-1460).
#include <cstring>
#include <memory>
#include <string>
#include <variant>

#include "Mila/Components.h"

using namespace Mila::Dnn;

namespace
{
    thread_local std::string g_lin_err;
    using Bf16 = Linear<DeviceType::Rocm, TensorDataType::BF16, Quant::Weight::NoWeightQuant>;
    using Fp8 = Linear<DeviceType::Rocm, TensorDataType::BF16, Quant::Weight::PerChannelFp8<>>;
    using Fp4 = Linear<DeviceType::Rocm, TensorDataType::BF16, Quant::Weight::PerGroupFp4<128>>;
    using TensorType = Tensor<TensorDataType::BF16, Compute::RocmDeviceMemoryResource>;
    struct LinearRunner
    {
        std::unique_ptr<Compute::IExecutionContext> ctx;
        std::variant<std::unique_ptr<Bf16>, std::unique_ptr<Fp8>, std::unique_ptr<Fp4>> lin;
        dim_t K, N, max_rows;
        std::unique_ptr<TensorType> x;
    };
    template<typename F> int guarded( F&& f )
    {
        try { f(); return 0; }
        catch ( const std::invalid_argument& e ) { g_lin_err = std::string( "invalid_argument: " ) + e.what(); return MILA_E_INVALID_ARGUMENT; }
        catch ( const std::logic_error& e ) { g_lin_err = std::string( "logic_error: " ) + e.what(); return MILA_E_UNSUPPORTED; }
        catch ( const std::exception& e ) { g_lin_err = e.what(); return MILA_E_RUNTIME; }
    }
    template<typename L> std::unique_ptr<L> build( LinearRunner& r, bool bias )
    {
        auto lin = std::make_unique<L>( "lin", LinearConfig( r.K, r.N ).withBias( bias ) );
        lin->setExecutionContext( r.ctx.get() );
        lin->build( BuildContext( shape_t{ 1, r.max_rows, r.K }, RuntimeMode::Inference ) );
        return lin;
    }
}

extern "C" {
#define HOST_API __attribute__((visibility("default")))
HOST_API const char* mila_linear_last_error( void ) { return g_lin_err.c_str(); }

/// policy 0 NoWeightQuant / 1 PerChannelFp8<> / 2 PerGroupFp4<128>; built for up to max_rows input rows
HOST_API void* mila_linear_create( int policy, int64_t K, int64_t N, int64_t max_rows, int bias, int device )
{
    LinearRunner* out = nullptr;
    int rc = guarded( [&]
    {
        auto r = std::make_unique<LinearRunner>();
        r->ctx = Compute::createExecutionContext( Compute::Device::Rocm( device ) );
        r->K = K; r->N = N; r->max_rows = max_rows;
        if ( policy == 0 ) r->lin = build<Bf16>( *r, bias != 0 );
        else if ( policy == 1 ) r->lin = build<Fp8>( *r, bias != 0 );
        else if ( policy == 2 ) r->lin = build<Fp4>( *r, bias != 0 );
        else throw std::invalid_argument( "mila_linear_create: policy must be 0, 1 or 2" );
        r->x = std::make_unique<TensorType>( r->ctx->getDeviceId(), shape_t{ 1, max_rows, K } );
        out = r.release();
    } );
    return rc == 0 ? out : nullptr;
}
HOST_API void mila_linear_destroy( void* h ) { delete static_cast<LinearRunner*>( h ); }
/// Linear::loadParameter( name, host blob ): "weight" (bf16 [N, K]: quantized on load under a quantized policy; or the policy's storage form), "weight_scale", "bias"
HOST_API int mila_linear_load( void* h, const char* name, const void* blob, int64_t bytes )
{
    auto* r = static_cast<LinearRunner*>( h );
    return guarded( [&] { std::visit( [&]( auto& l ) { l->loadParameter( name, blob, static_cast<size_t>( bytes ) ); }, r->lin ); } );
}
/// the stored weight (storage form) and scales back on the host: weight_bytes / scale_bytes receive the sizes; NULL buffers = size query
HOST_API int mila_linear_read( void* h, void* weight_out, int64_t* weight_bytes, void* scale_out, int64_t* scale_bytes )
{
    auto* r = static_cast<LinearRunner*>( h );
    return guarded( [&]
    {
        std::visit( [&]( auto& l )
        {
            auto* ctx = Compute::cast_context<DeviceType::Rocm>( r->ctx.get() );
            auto& w = l->getWeight();
            if ( weight_bytes ) *weight_bytes = static_cast<int64_t>( w.sizeInBytes() );
            if ( weight_out ) copyToHost( weight_out, w, w.sizeInBytes(), ctx );
            auto* s = l->getWeightScale();
            if ( scale_bytes ) *scale_bytes = s ? static_cast<int64_t>( s->sizeInBytes() ) : 0;
            if ( scale_out && s ) copyToHost( scale_out, *s, s->sizeInBytes(), ctx );
        }, r->lin );
    } );
}
/// quantized policies: RocmLinearOp::setFp8ActivationPrefill / setResidentPrefillWeights
HOST_API int mila_linear_set( void* h, int fp8_activation_prefill, int resident )
{
    auto* r = static_cast<LinearRunner*>( h );
    return guarded( [&]
    {
        std::visit( [&]( auto& l )
        {
            if ( fp8_activation_prefill >= 0 ) l->getOperation().setFp8ActivationPrefill( fp8_activation_prefill != 0 );
            if ( resident >= 0 ) l->getOperation().setResidentPrefillWeights( resident != 0 );
        }, r->lin );
    } );
}
/// the tying contract's rejections (Linear.Cuda.cpp:618-641): which = 0: installSharedWeight( nullptr ); 1: installSharedWeight( nullptr, nullptr ) on an UNBUILT
/// Linear of `policy`; returns MILA_E_UNSUPPORTED for std::logic_error, MILA_E_INVALID_ARGUMENT for std::invalid_argument (the null weight of an accepted overload)
HOST_API int mila_linear_install_shared_probe( int policy, int which )
{
    return guarded( [&]
    {
        auto probe = [&]( auto lin )
        {
            using L = decltype( lin );
            if ( which == 0 ) lin.installSharedWeight( std::shared_ptr<typename L::WeightTensorType>() );
            else lin.installSharedWeight( std::shared_ptr<typename L::WeightTensorType>(), std::shared_ptr<typename L::WeightScaleTensorType>() );
        };
        const LinearConfig cfg = LinearConfig( 256, 128 ).withBias( false );
        if ( policy == 0 ) probe( Bf16( "linear", cfg ) );
        else if ( policy == 1 ) probe( Fp8( "linear_quantized", cfg ) );
        else probe( Fp4( "linear_quantized", cfg ) );
    } );
}

// ---- RocmSamplingOp (Sampling.Cuda.cpp:40-150: makeOp / sample / sampleEnqueued) ----
namespace
{
    struct SamplerRunner
    {
        std::unique_ptr<Compute::IExecutionContext> ctx;
        std::unique_ptr<Compute::RocmSamplingOp> op;
        std::unique_ptr<Compute::RocmSamplingOp::LogitsTensor> logits;
        std::unique_ptr<Compute::RocmSamplingOp::TokenTensor> token;
        dim_t vocab;
    };
}
HOST_API void* mila_sampler_create( int64_t vocab, float softcap, int device )
{
    SamplerRunner* out = nullptr;
    int rc = guarded( [&]
    {
        auto r = std::make_unique<SamplerRunner>();
        r->ctx = Compute::createExecutionContext( Compute::Device::Rocm( device ) );
        r->op = std::make_unique<Compute::RocmSamplingOp>( r->ctx.get(), Compute::SamplingOpConfig{ vocab, softcap } );
        r->logits = std::make_unique<Compute::RocmSamplingOp::LogitsTensor>( r->ctx->getDeviceId(), shape_t{ 1, 1, vocab } );
        r->token = std::make_unique<Compute::RocmSamplingOp::TokenTensor>( r->ctx->getDeviceId(), shape_t{ 1, 1 } );
        r->vocab = vocab;
        out = r.release();
    } );
    return rc == 0 ? out : nullptr;
}
HOST_API void mila_sampler_destroy( void* h ) { delete static_cast<SamplerRunner*>( h ); }
/// copy host logits to the device ON THE CONTEXT STREAM, without a host synchronize (the enqueued path's ordering contract, Sampling.Cuda.cpp:478-500)
HOST_API int mila_sampler_set_logits( void* h, const float* host_logits )
{
    auto* r = static_cast<SamplerRunner*>( h );
    return guarded( [&] { Compute::rocmCheck( mila_cdna4_memcpy_h2d( r->logits->rawData(), host_logits, static_cast<size_t>( r->vocab ) * 4, Compute::cast_context<DeviceType::Rocm>( r->ctx.get() )->getStream() ) ); } );
}
/// enqueued 0: forward() + a synchronous readback; 1: enqueueForward() + awaitToken(); 2: awaitToken() alone (the caller-bug case)
HOST_API int mila_sampler_sample( void* h, int enqueued, float temperature, int top_k, float top_p, float rnd, int32_t* token_out )
{
    auto* r = static_cast<SamplerRunner*>( h );
    return guarded( [&]
    {
        const Compute::SamplingParams sp{ temperature, top_k, top_p };
        auto* ctx = Compute::cast_context<DeviceType::Rocm>( r->ctx.get() );
        if ( enqueued == 2 ) { *token_out = r->op->awaitToken(); return; }
        if ( enqueued == 1 ) { r->op->enqueueForward( *r->logits, *r->token, sp, rnd ); *token_out = r->op->awaitToken(); return; }
        r->op->forward( *r->logits, *r->token, sp, rnd );
        Compute::rocmCheck( mila_cdna4_memcpy_d2h( token_out, r->token->rawData(), 4, ctx->getStream() ) );
        ctx->synchronize();
    } );
}

// ---- the activation tap (Operations.h: ActivationTap): begin -> run any prefill on this thread -> count / get -> end ----
namespace { Compute::ActivationTap g_tap; }
HOST_API void mila_linear_tap_begin( void ) { g_tap.records.clear(); Compute::activationTap() = &g_tap; }
HOST_API void mila_linear_tap_end( void ) { Compute::activationTap() = nullptr; }
HOST_API int64_t mila_linear_tap_count( void ) { return static_cast<int64_t>( g_tap.records.size() ); }
/// dims[3] = {M, K, N}; x8_out [M, K] / ts_out [M] may be NULL (size query)
HOST_API int mila_linear_tap_get( int64_t i, int32_t* dims, uint8_t* x8_out, float* ts_out )
{
    return guarded( [&]
    {
        if ( i < 0 || i >= static_cast<int64_t>( g_tap.records.size() ) ) throw std::invalid_argument( "mila_linear_tap_get: no such record" );
        const auto& r = g_tap.records[ static_cast<size_t>( i ) ];
        if ( dims ) { dims[ 0 ] = r.M; dims[ 1 ] = r.K; dims[ 2 ] = r.N; }
        if ( x8_out ) std::memcpy( x8_out, r.x8.data(), r.x8.size() );
        if ( ts_out ) std::memcpy( ts_out, r.ts.data(), r.ts.size() * 4 );
    } );
}

/// Linear::forward on M host rows (bf16 bits) -> M x N host rows
HOST_API int mila_linear_forward( void* h, int64_t M, const uint16_t* x_host, uint16_t* y_host )
{
    auto* r = static_cast<LinearRunner*>( h );
    return guarded( [&]
    {
        if ( M <= 0 || M > r->max_rows ) throw std::invalid_argument( "mila_linear_forward: row count outside (0, max_rows]" );
        auto* ctx = Compute::cast_context<DeviceType::Rocm>( r->ctx.get() );
        Compute::rocmCheck( mila_cdna4_memcpy_h2d( r->x->rawData(), x_host, static_cast<size_t>( M * r->K ) * 2, ctx->getStream() ) );
        auto xin = r->x->view( shape_t{ 1, M, r->K } );
        std::visit( [&]( auto& l )
        {
            auto& y = l->forward( xin );
            ctx->synchronize();
            copyToHost( y_host, y, static_cast<size_t>( M * r->N ) * 2, ctx );
        }, r->lin );
    } );
}
}
