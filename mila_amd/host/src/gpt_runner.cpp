// C entry points for the GPT-2 host mirror (BASELINE.json configs 1-2): GptTransformerT<BF16> (config 2) and GptTransformerT<FP32> (config 1's model on the device).
#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <variant>

#include "Mila/Gpt.h"

using namespace Mila::Dnn;

namespace
{
    thread_local std::string g_err;
    using TokenTensor = GptTransformer::TokenTensor;
    struct GptRunner
    {
        std::variant<std::unique_ptr<GptTransformer>, std::unique_ptr<GptTransformerFp32>> model;
        std::unique_ptr<TokenTensor> tokens;
        dim_t B, T;
    };
    template<typename F> int guarded( F&& f )
    {
        try { f(); return 0; }
        catch ( const std::invalid_argument& e ) { g_err = std::string( "invalid_argument: " ) + e.what(); return MILA_E_INVALID_ARGUMENT; }
        catch ( const std::exception& e ) { g_err = e.what(); return MILA_E_RUNTIME; }
    }
    template<typename TModel>
    GptRunner* make_runner( int64_t vocab, int64_t max_seq, int64_t C, int64_t L, int64_t NH, int64_t B, int64_t T )
    {
        GptConfig cfg;
        cfg.vocab_size = vocab; cfg.max_seq_len = max_seq; cfg.embedding_dim = C; cfg.num_layers = L; cfg.num_heads = NH;
        auto rr = std::make_unique<GptRunner>();
        auto m = std::make_unique<TModel>( cfg, B, T );
        rr->tokens = std::make_unique<TokenTensor>( m->context()->getDeviceId(), shape_t{ B, T } );
        rr->model = std::move( m );
        rr->B = B; rr->T = T;
        return rr.release();
    }
}

extern "C" {
#define HOST_API __attribute__((visibility("default")))

HOST_API const char* mila_gpt_last_error( void ) { return g_err.c_str(); }

/// precision 0 = BF16 (parameters and logits as bf16 bit patterns), 1 = FP32 (float parameters and logits)
HOST_API void* mila_gpt_create_p( int precision, int64_t vocab, int64_t max_seq, int64_t C, int64_t L, int64_t NH, int64_t B, int64_t T )
{
    GptRunner* r = nullptr;
    int rc = guarded( [&]
    {
        if ( precision == 0 ) r = make_runner<GptTransformer>( vocab, max_seq, C, L, NH, B, T );
        else if ( precision == 1 ) r = make_runner<GptTransformerFp32>( vocab, max_seq, C, L, NH, B, T );
        else throw std::invalid_argument( "mila_gpt_create_p: precision must be 0 (BF16) or 1 (FP32)" );
    } );
    return rc == 0 ? r : nullptr;
}
HOST_API void* mila_gpt_create( int64_t vocab, int64_t max_seq, int64_t C, int64_t L, int64_t NH, int64_t B, int64_t T ) { return mila_gpt_create_p( 0, vocab, max_seq, C, L, NH, B, T ); }
HOST_API void mila_gpt_destroy( void* h ) { delete static_cast<GptRunner*>( h ); }
HOST_API int64_t mila_gpt_parameter_count( void* h )
{
    return std::visit( []( auto& m ) { return static_cast<int64_t>( m->parameterCount() ); }, static_cast<GptRunner*>( h )->model );
}
/// host blob in the model's precision (bf16 bit patterns, or floats)
HOST_API int mila_gpt_load_parameter( void* h, int64_t index, const void* host_blob, int64_t bytes )
{
    return guarded( [&] { std::visit( [&]( auto& m ) { m->loadParameter( static_cast<size_t>( index ), host_blob, static_cast<size_t>( bytes ) ); }, static_cast<GptRunner*>( h )->model ); } );
}
/// out[0..1] = getRequiredMemory(): device parameter / state bytes; out[2..3] = getMemoryStats() likewise
HOST_API int mila_gpt_memory_stats( void* h, double* out )
{
    return guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            const Mila::Dnn::MemoryStats req = m->getRequiredMemory(), act = m->getMemoryStats();
            out[ 0 ] = static_cast<double>( req.device_parameter_bytes ); out[ 1 ] = static_cast<double>( req.device_state_bytes );
            out[ 2 ] = static_cast<double>( act.device_parameter_bytes ); out[ 3 ] = static_cast<double>( act.device_state_bytes );
        }, static_cast<GptRunner*>( h )->model );
    } );
}
/// component names in construction order, '\n'-separated; returns the bytes needed (with the terminator)
HOST_API int64_t mila_gpt_component_names( void* h, char* buf, int64_t cap )
{
    std::string out;
    try { std::visit( [&]( auto& m ) { for ( const auto& n : m->componentNames() ) out += n + "\n"; }, static_cast<GptRunner*>( h )->model ); }
    catch ( const std::exception& e ) { g_err = e.what(); return -1; }
    if ( buf && cap > 0 ) { const size_t n = std::min<size_t>( out.size(), static_cast<size_t>( cap - 1 ) ); std::memcpy( buf, out.data(), n ); buf[ n ] = 0; }
    return static_cast<int64_t>( out.size() + 1 );
}

/// GptTransformer::prefill: host tokens [B, Tp] -> host logits [B, V] (the model's precision) of the last position; fills every block's KV cache
HOST_API int mila_gpt_prefill( void* h, const int32_t* host_tokens, int64_t Tp, void* host_logits )
{
    auto* r = static_cast<GptRunner*>( h );
    return guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            if ( Tp <= 0 || Tp > r->T ) throw std::invalid_argument( "mila_gpt_prefill: prompt length outside (0, built T]" );
            TokenTensor toks( ctx->getDeviceId(), shape_t{ r->B, Tp } );
            Compute::rocmCheck( mila_cdna4_memcpy_h2d( toks.data(), host_tokens, static_cast<size_t>( r->B * Tp ) * 4, ctx->getStream() ) );
            auto& logits = m->prefill( toks );
            ctx->synchronize();
            if ( m->indexError() ) throw std::invalid_argument( "mila_gpt_prefill: token index outside the vocabulary" );
            if ( host_logits ) copyToHost( host_logits, logits, logits.sizeInBytes(), ctx );
        }, r->model );
    } );
}
/// GptTransformer::decode: host tokens [B] at absolute `position` -> host logits [B, V]
HOST_API int mila_gpt_decode( void* h, const int32_t* host_tokens, int64_t position, void* host_logits )
{
    auto* r = static_cast<GptRunner*>( h );
    return guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            TokenTensor toks( ctx->getDeviceId(), shape_t{ r->B, 1 } );
            Compute::rocmCheck( mila_cdna4_memcpy_h2d( toks.data(), host_tokens, static_cast<size_t>( r->B ) * 4, ctx->getStream() ) );
            auto& logits = m->decode( toks, position );
            ctx->synchronize();
            if ( host_logits ) copyToHost( host_logits, logits, logits.sizeInBytes(), ctx );
        }, r->model );
    } );
}

/// tokens [B, T] host int32 -> logits [B, T, V] on the host (the model's precision; NULL = timing only); returns 0, a negative error, or the positive 1-based flat index
/// of an out-of-range token
HOST_API int mila_gpt_forward( void* h, const int32_t* host_tokens, void* host_logits, double* ms )
{
    auto* r = static_cast<GptRunner*>( h );
    int bad = 0;
    int rc = guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            Compute::rocmCheck( mila_cdna4_memcpy_h2d( r->tokens->data(), host_tokens, static_cast<size_t>( r->B * r->T ) * 4, ctx->getStream() ) );
            ctx->synchronize();
            hipEvent_t e0, e1;
            (void)hipEventCreate( &e0 ); (void)hipEventCreate( &e1 );
            (void)hipEventRecord( e0, reinterpret_cast<hipStream_t>( ctx->getStream() ) );
            auto& logits = m->forward( *r->tokens );
            (void)hipEventRecord( e1, reinterpret_cast<hipStream_t>( ctx->getStream() ) );
            ctx->synchronize();
            float t = 0; (void)hipEventElapsedTime( &t, e0, e1 );
            (void)hipEventDestroy( e0 ); (void)hipEventDestroy( e1 );
            if ( ms ) *ms = t;
            bad = m->indexError();
            if ( host_logits ) copyToHost( host_logits, logits, logits.sizeInBytes(), ctx );
        }, r->model );
    } );
    return rc ? rc : bad;
}
}
