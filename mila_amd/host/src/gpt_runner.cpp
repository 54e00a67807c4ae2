// C entry points for the GPT-2 host mirror (BASELINE.json configs 1-2).
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>

#include "Mila/Gpt.h"

using namespace Mila::Dnn;

namespace
{
    thread_local std::string g_err;
    struct GptRunner
    {
        std::unique_ptr<GptTransformer> model;
        std::unique_ptr<GptTransformer::TokenTensor> tokens;
        dim_t B, T;
    };
    template<typename F> int guarded( F&& f )
    {
        try { f(); return 0; }
        catch ( const std::invalid_argument& e ) { g_err = std::string( "invalid_argument: " ) + e.what(); return MILA_E_INVALID_ARGUMENT; }
        catch ( const std::exception& e ) { g_err = e.what(); return MILA_E_RUNTIME; }
    }
}

extern "C" {
#define HOST_API __attribute__((visibility("default")))

HOST_API const char* mila_gpt_last_error( void ) { return g_err.c_str(); }

HOST_API void* mila_gpt_create( int64_t vocab, int64_t max_seq, int64_t C, int64_t L, int64_t NH, int64_t B, int64_t T )
{
    GptRunner* r = nullptr;
    int rc = guarded( [&]
    {
        GptConfig cfg;
        cfg.vocab_size = vocab; cfg.max_seq_len = max_seq; cfg.embedding_dim = C; cfg.num_layers = L; cfg.num_heads = NH;
        auto rr = std::make_unique<GptRunner>();
        rr->model = std::make_unique<GptTransformer>( cfg, B, T );
        rr->tokens = std::make_unique<GptTransformer::TokenTensor>( rr->model->context()->getDeviceId(), shape_t{ B, T } );
        rr->B = B; rr->T = T;
        r = rr.release();
    } );
    return rc == 0 ? r : nullptr;
}
HOST_API void mila_gpt_destroy( void* h ) { delete static_cast<GptRunner*>( h ); }
HOST_API int64_t mila_gpt_parameter_count( void* h ) { return static_cast<int64_t>( static_cast<GptRunner*>( h )->model->parameterCount() ); }
HOST_API int mila_gpt_load_parameter( void* h, int64_t index, const void* host_bf16, int64_t bytes )
{
    return guarded( [&] { static_cast<GptRunner*>( h )->model->loadParameter( static_cast<size_t>( index ), host_bf16, static_cast<size_t>( bytes ) ); } );
}
/// out[0..1] = getRequiredMemory(): device parameter / state bytes; out[2..3] = getMemoryStats() likewise
HOST_API int mila_gpt_memory_stats( void* h, double* out )
{
    return guarded( [&]
    {
        auto& m = *static_cast<GptRunner*>( h )->model;
        const Mila::Dnn::MemoryStats req = m.getRequiredMemory(), act = m.getMemoryStats();
        out[ 0 ] = static_cast<double>( req.device_parameter_bytes ); out[ 1 ] = static_cast<double>( req.device_state_bytes );
        out[ 2 ] = static_cast<double>( act.device_parameter_bytes ); out[ 3 ] = static_cast<double>( act.device_state_bytes );
    } );
}
/// tokens [B,T] host int32 -> logits [B,T,V] host bf16 bits; returns 0, or a positive 1-based index of an out-of-range token
/// component names in construction order, '\n'-separated; returns the bytes needed (with the terminator)
HOST_API int64_t mila_gpt_component_names( void* h, char* buf, int64_t cap )
{
    std::string out;
    try { for ( const auto& n : static_cast<GptRunner*>( h )->model->componentNames() ) out += n + "\n"; }
    catch ( const std::exception& e ) { g_err = e.what(); return -1; }
    if ( buf && cap > 0 ) { const size_t n = std::min<size_t>( out.size(), static_cast<size_t>( cap - 1 ) ); std::memcpy( buf, out.data(), n ); buf[ n ] = 0; }
    return static_cast<int64_t>( out.size() + 1 );
}

/// GptTransformer::prefill: host tokens [B, Tp] -> host logits [B, V] (bf16 bits) of the last position; fills every block's KV cache
HOST_API int mila_gpt_prefill( void* h, const int32_t* host_tokens, int64_t Tp, uint16_t* host_logits )
{
    auto* r = static_cast<GptRunner*>( h );
    return guarded( [&]
    {
        auto* ctx = r->model->context();
        if ( Tp <= 0 || Tp > r->T ) throw std::invalid_argument( "mila_gpt_prefill: prompt length outside (0, built T]" );
        GptTransformer::TokenTensor toks( ctx->getDeviceId(), shape_t{ r->B, Tp } );
        Compute::rocmCheck( mila_cdna4_memcpy_h2d( toks.data(), host_tokens, static_cast<size_t>( r->B * Tp ) * 4, ctx->getStream() ) );
        auto& logits = r->model->prefill( toks );
        ctx->synchronize();
        if ( r->model->indexError() ) throw std::invalid_argument( "mila_gpt_prefill: token index outside the vocabulary" );
        if ( host_logits ) copyToHost( host_logits, logits, logits.sizeInBytes(), ctx );
    } );
}
/// GptTransformer::decode: host tokens [B] at absolute `position` -> host logits [B, V] (bf16 bits)
HOST_API int mila_gpt_decode( void* h, const int32_t* host_tokens, int64_t position, uint16_t* host_logits )
{
    auto* r = static_cast<GptRunner*>( h );
    return guarded( [&]
    {
        auto* ctx = r->model->context();
        GptTransformer::TokenTensor toks( ctx->getDeviceId(), shape_t{ r->B, 1 } );
        Compute::rocmCheck( mila_cdna4_memcpy_h2d( toks.data(), host_tokens, static_cast<size_t>( r->B ) * 4, ctx->getStream() ) );
        auto& logits = r->model->decode( toks, position );
        ctx->synchronize();
        if ( host_logits ) copyToHost( host_logits, logits, logits.sizeInBytes(), ctx );
    } );
}

HOST_API int mila_gpt_forward( void* h, const int32_t* host_tokens, uint16_t* host_logits, double* ms )
{
    auto* r = static_cast<GptRunner*>( h );
    int bad = 0;
    int rc = guarded( [&]
    {
        auto* ctx = r->model->context();
        Compute::rocmCheck( mila_cdna4_memcpy_h2d( r->tokens->data(), host_tokens, static_cast<size_t>( r->B * r->T ) * 4, ctx->getStream() ) );
        ctx->synchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate( &e0 ); (void)hipEventCreate( &e1 );
        (void)hipEventRecord( e0, reinterpret_cast<hipStream_t>( ctx->getStream() ) );
        auto& logits = r->model->forward( *r->tokens );
        (void)hipEventRecord( e1, reinterpret_cast<hipStream_t>( ctx->getStream() ) );
        ctx->synchronize();
        float t = 0; (void)hipEventElapsedTime( &t, e0, e1 );
        (void)hipEventDestroy( e0 ); (void)hipEventDestroy( e1 );
        if ( ms ) *ms = t;
        bad = r->model->indexError();
        if ( host_logits ) copyToHost( host_logits, logits, logits.sizeInBytes(), ctx );
    } );
    return rc ? rc : bad;
}
}
