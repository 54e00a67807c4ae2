// C entry points over the C++ host mirror for bench.py / tests (plain pointers and sizes only).
// Instantiates GemmaTransformer<TWeightQuant> for the three weight policies of BASELINE.json
// configs 3-5 and exposes build / prefill / decode / timing.
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <random>
#include <span>
#include <string>
#include <variant>

#include "Mila/GemmaModel.h"

using namespace Mila::Dnn;
using Quant::Weight::NoWeightQuant;
using Quant::Weight::PerChannelFp8;
using Quant::Weight::PerGroupFp4;

namespace
{
    thread_local std::string g_err;

    struct Runner
    {
        std::variant<std::unique_ptr<GemmaTransformer<NoWeightQuant>>, std::unique_ptr<GemmaTransformer<PerChannelFp8<>>>,
                     std::unique_ptr<GemmaTransformer<PerGroupFp4<128>>>> model;
        std::unique_ptr<Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource>> tokens;
        dim_t max_prefill{ 1 };
    };

    template<typename F> int guarded( F&& f )
    {
        try { f(); return 0; }
        catch ( const std::invalid_argument& e ) { g_err = std::string( "invalid_argument: " ) + e.what(); return MILA_E_INVALID_ARGUMENT; }
        catch ( const std::exception& e ) { g_err = e.what(); return MILA_E_RUNTIME; }
    }
}

// Gemma.Cuda.cpp:456-480 (KvPolicy_RoutesBoundedRingToLocalLayersOnly), the compile-time half: the sliding-window KV policy reaches the LOCAL block type only; global
// (full-attention) blocks are never bounded, whatever the policy.  (The byte footprint of the routing: tests/test_reference_scenarios_r4_gpu.py.)
namespace
{
    using ProbeNet = Mila::Dnn::GemmaTransformer<Mila::Dnn::Quant::Weight::NoWeightQuant>;
    static_assert( std::is_same_v<ProbeNet::BoundedLocalBlockType::AttentionType::OpType, Mila::Dnn::Compute::RocmGqaOp<true>>, "SlidingWindowKvCache must reach the local (sliding) layers" );
    static_assert( std::is_same_v<ProbeNet::LocalBlockType::AttentionType::OpType, Mila::Dnn::Compute::RocmGqaOp<false>>, "the default policy keeps local layers unbounded" );
    static_assert( std::is_same_v<ProbeNet::GlobalBlockType::AttentionType::OpType, Mila::Dnn::Compute::RocmGqaOp<false>>, "global (full-attention) layers must never be bounded" );
}

extern "C" {

#define HOST_API __attribute__((visibility("default")))

struct mila_gemma_config
{
    int64_t vocab_size, embedding_dim, num_layers, num_heads, num_kv_heads, head_dim, hidden_dim, global_head_dim,
            num_global_kv_heads, window, sliding_window_pattern, global_rotary_dim;
    int64_t bounded_local_kv;   ///< != 0: SlidingWindowKvCache (ring of window + chunk - 1 rows) on the sliding-window layers
};

HOST_API const char* mila_host_last_error( void ) { return g_err.c_str(); }

/// policy: 0 NoWeightQuant (bf16), 1 PerChannelFp8<>, 2 PerGroupFp4<128>.  cfg == NULL -> Gemma-4 12B.
/// device: HIP device ordinal of this replica (one process per GPU: the launcher's LOCAL_RANK)
HOST_API void* mila_gemma_create( int policy, const mila_gemma_config* c, int64_t max_seq, int64_t max_prefill, uint64_t seed, int device )
{
    Runner* r = nullptr;
    int rc = guarded( [&]
    {
        GemmaConfig cfg;
        if ( c )
        {
            cfg.vocab_size = c->vocab_size; cfg.embedding_dim = c->embedding_dim; cfg.num_layers = c->num_layers; cfg.num_heads = c->num_heads;
            cfg.num_kv_heads = c->num_kv_heads; cfg.head_dim = c->head_dim; cfg.hidden_dim = c->hidden_dim; cfg.global_head_dim = c->global_head_dim;
            cfg.num_global_kv_heads = c->num_global_kv_heads; cfg.window = c->window; cfg.sliding_window_pattern = c->sliding_window_pattern;
            cfg.global_rotary_dim = c->global_rotary_dim;
            cfg.bounded_local_kv = c->bounded_local_kv != 0;
        }
        auto rr = std::make_unique<Runner>();
        rr->max_prefill = max_prefill;
        switch ( policy )
        {
            case 0: rr->model = std::make_unique<GemmaTransformer<NoWeightQuant>>( cfg, max_seq, max_prefill, Compute::Device::Rocm( device ) ); break;
            case 1: rr->model = std::make_unique<GemmaTransformer<PerChannelFp8<>>>( cfg, max_seq, max_prefill, Compute::Device::Rocm( device ) ); break;
            case 2: rr->model = std::make_unique<GemmaTransformer<PerGroupFp4<128>>>( cfg, max_seq, max_prefill, Compute::Device::Rocm( device ) ); break;
            default: throw std::invalid_argument( "unknown weight policy" );
        }
        std::visit( [&]( auto& m )
        {
            m->initSynthetic( seed );
            rr->tokens = std::make_unique<Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource>>(
                m->context()->getDeviceId(), shape_t{ std::max<dim_t>( max_prefill, 1 ) } );
        }, rr->model );
        r = rr.release();
    } );
    return rc == 0 ? r : nullptr;
}

/// regenerate the synthetic parameters under a profile: p = { linear_gain, qk_norm_center, post_norm_center, layer_scalar, table_gain }
HOST_API int mila_gemma_init_synthetic( void* h, uint64_t seed, const float* p )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        if ( !p ) throw std::invalid_argument( "init_synthetic: null profile" );
        std::visit( [&]( auto& m )
        {
            typename std::remove_reference_t<decltype( *m )>::SyntheticProfile pr;
            pr.linear_gain = p[ 0 ]; pr.qk_norm_center = p[ 1 ]; pr.post_norm_center = p[ 2 ]; pr.layer_scalar = p[ 3 ]; pr.table_gain = p[ 4 ];
            m->initSynthetic( seed, pr );
        }, r->model );
    } );
}

HOST_API void mila_gemma_destroy( void* h ) { delete static_cast<Runner*>( h ); }

static void upload_tokens( Runner* r, const int32_t* host_tokens, int64_t n )
{
    std::visit( [&]( auto& m )
    {
        Compute::rocmCheck( mila_cdna4_memcpy_h2d( r->tokens->data(), host_tokens, static_cast<size_t>( n ) * 4, m->context()->getStream() ) );
        m->context()->synchronize();
    }, r->model );
}

static void download_logits( Runner* r, float* host_logits )
{
    if ( !host_logits ) return;
    std::visit( [&]( auto& m )
    {
        auto& lg = m->logits();
        Compute::rocmCheck( mila_cdna4_memcpy_d2h( host_logits, lg.data(), lg.sizeInBytes(), m->context()->getStream() ) );
        m->context()->synchronize();
    }, r->model );
}

HOST_API int mila_gemma_prefill( void* h, const int32_t* host_tokens, int64_t T, int64_t position_offset, float* host_logits )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        if ( T <= 0 || T > r->max_prefill ) throw std::invalid_argument( "GemmaTransformer::prefill: chunk length out of range" );
        upload_tokens( r, host_tokens, T );
        std::visit( [&]( auto& m ) { m->prefill( *r->tokens, T, position_offset ); m->context()->synchronize(); }, r->model );
        download_logits( r, host_logits );
    } );
}

/// on != 0 (default): prefill runs the fused glue kernels when the configuration fits; 0: one launch per reference op
HOST_API int mila_gemma_set_fused_prefill( void* h, int on )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { std::visit( [&]( auto& m ) { m->setFusedPrefill( on != 0 ); }, r->model ); } );
}
/// on != 0 (default): layers with a small split-partial set run the attention combine inside o_proj's prologue
/// GemmaTransformer::rewindKvCache( position ): *ok = 1 when every block accepted
HOST_API int mila_gemma_rewind( void* h, int64_t position, int* ok )
{
    return guarded( [&] { std::visit( [&]( auto& m ) { *ok = m->rewindKvCache( position ) ? 1 : 0; }, static_cast<Runner*>( h )->model ); } );
}

/// GemmaTransformer::prefillFrom( tokens[0 .. T), offset ): positions [0, offset) stay resident, the tail is prefilled; logits of the last position
HOST_API int mila_gemma_prefill_from( void* h, const int32_t* host_tokens, int64_t T, int64_t offset, float* host_logits )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            if ( T <= 0 ) throw std::invalid_argument( "prefill_from: empty prompt" );
            Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource> toks( m->context()->getDeviceId(), shape_t{ 1, T } );
            Compute::rocmCheck( mila_cdna4_memcpy_h2d( toks.rawData(), host_tokens, static_cast<size_t>( T ) * 4, m->context()->getStream() ) );
            m->prefillFrom( toks, T, offset );
            m->context()->synchronize();
        }, r->model );
        download_logits( r, host_logits );
    } );
}

HOST_API int mila_gemma_set_prefill_overlap( void* h, int on )
{
    return guarded( [&] { std::visit( [&]( auto& m ) { m->setPrefillOverlap( on != 0 ); }, static_cast<Runner*>( h )->model ); } );
}

/// every parameter of the model in its storage form -> a SafeTensors file / back (component paths as tensor names)
HOST_API int mila_gemma_save_safetensors( void* h, const char* path )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { if ( !path ) throw std::invalid_argument( "save_safetensors: null path" ); std::visit( [&]( auto& m ) { m->saveSafeTensors( path ); }, r->model ); } );
}
HOST_API int mila_gemma_load_safetensors( void* h, const char* path )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { if ( !path ) throw std::invalid_argument( "load_safetensors: null path" ); std::visit( [&]( auto& m ) { m->loadSafeTensors( path ); }, r->model ); } );
}
/// the same tensors as a MILA .bin container / load from either container (sniffed by the leading magic, PretrainedReader.ixx:283-293)
HOST_API int mila_gemma_save_milabin( void* h, const char* path )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { if ( !path ) throw std::invalid_argument( "save_milabin: null path" ); std::visit( [&]( auto& m ) { m->saveMilaBin( path ); }, r->model ); } );
}
HOST_API int mila_gemma_load_pretrained( void* h, const char* path )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { if ( !path ) throw std::invalid_argument( "load_pretrained: null path" ); std::visit( [&]( auto& m ) { m->loadPretrained( path ); }, r->model ); } );
}
/// host-only (no device): list either container as the PretrainedModelReader sees it -- "name dtype nbytes d0,d1,.." lines in ascending
/// file-offset order, then "# container=mila|safetensors", "# mila_quantization=..", "# mila_config=<json>" lines
HOST_API int64_t mila_pretrained_list( const char* path, char* out, int64_t cap )
{
    int64_t need = -1;
    const int rc = guarded( [&]
    {
        Mila::Dnn::Serialization::PretrainedModelReader rd( path ? path : "" );
        std::string t;
        for ( auto& e : rd.entries() )
        {
            t += e.name + " " + e.dtype + " " + std::to_string( e.nbytes() ) + " ";
            for ( size_t d = 0; d < e.shape.size(); ++d ) t += ( d ? "," : "" ) + std::to_string( e.shape[ d ] );
            t += "\n";
        }
        t += std::string( "# container=" ) + ( rd.isMilaContainer() ? "mila" : "safetensors" ) + "\n";
        t += "# mila_quantization=" + rd.getWeightQuantization() + "\n";
        t += "# mila_config=" + rd.metadataJSON() + "\n";
        need = (int64_t)t.size();
        if ( out && cap > 0 ) { const size_t n = std::min<size_t>( t.size(), (size_t)cap - 1 ); std::memcpy( out, t.data(), n ); out[ n ] = 0; }
    } );
    return rc ? rc : need;
}
/// host-only: rewrite either container as a MILA .bin (tensor order = ascending source offsets); metadata_json == NULL keeps the source's
HOST_API int mila_pretrained_to_milabin( const char* src, const char* dst, const char* metadata_json )
{
    return guarded( [&]
    {
        Mila::Dnn::Serialization::PretrainedModelReader rd( src ? src : "" );
        Mila::Dnn::Serialization::MilaBinWriter wr( dst ? dst : "" );
        for ( auto& e : rd.entries() ) wr.declareTensor( e.name, e.dtype, e.shape );
        wr.setMetadataJSON( metadata_json ? std::string( metadata_json ) : rd.metadataJSON() );
        wr.beginData();
        for ( auto& e : rd.entries() ) wr.writeTensorData( e.name, e.data, e.nbytes() );
        wr.close();
    } );
}
/// host-only: the value the metadata parser extracts for `key` from a JSON text, round-tripped through toMetadataJSON (parser check)
HOST_API int64_t mila_pretrained_metadata_roundtrip( const char* json, char* out, int64_t cap )
{
    int64_t need = -1;
    const int rc = guarded( [&]
    {
        const std::string t = Mila::Dnn::Serialization::toMetadataJSON( Mila::Dnn::Serialization::parseMetadataJSON( json ? json : "" ) );
        need = (int64_t)t.size();
        if ( out && cap > 0 ) { const size_t n = std::min<size_t>( t.size(), (size_t)cap - 1 ); std::memcpy( out, t.data(), n ); out[ n ] = 0; }
    } );
    return rc ? rc : need;
}
/// host-only container checks (no device): list a file as "name dtype nbytes d0,d1,..\n" lines + "# key=value" metadata lines into out
/// (returns the length needed, or < 0 with mila_gemma_last_error set); copy src -> dst tensor by tensor through the reader and the writer
HOST_API int64_t mila_safetensors_list( const char* path, char* out, int64_t cap )
{
    int64_t need = -1;
    const int rc = guarded( [&]
    {
        Mila::Dnn::Serialization::SafeTensorsReader rd( path ? path : "" );
        std::string t;
        for ( auto& e : rd.entries() )
        {
            t += e.name + " " + e.dtype + " " + std::to_string( e.nbytes() ) + " ";
            for ( size_t d = 0; d < e.shape.size(); ++d ) t += ( d ? "," : "" ) + std::to_string( e.shape[ d ] );
            t += "\n";
        }
        for ( auto& [ k, v ] : rd.metadata() ) t += "# " + k + "=" + v + "\n";
        need = (int64_t)t.size();
        if ( out && cap > 0 ) { const size_t n = std::min<size_t>( t.size(), (size_t)cap - 1 ); std::memcpy( out, t.data(), n ); out[ n ] = 0; }
    } );
    return rc ? rc : need;
}
HOST_API int mila_safetensors_copy( const char* src, const char* dst )
{
    return guarded( [&]
    {
        Mila::Dnn::Serialization::SafeTensorsReader rd( src ? src : "" );
        Mila::Dnn::Serialization::SafeTensorsWriter wr( dst ? dst : "" );
        for ( auto& e : rd.entries() ) wr.declareTensor( e.name, e.dtype, e.shape );
        for ( auto& [ k, v ] : rd.metadata() ) wr.setMetadata( k, v );
        wr.beginData();
        for ( auto& e : rd.entries() ) wr.writeTensorData( e.name, e.data, e.nbytes() );
        wr.close();
    } );
}
/// on != 0 (default): quantized policies keep their prefill staging resident; 0: re-stage on every forward (the reference's way)
HOST_API int mila_gemma_set_resident_prefill_weights( void* h, int on )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { std::visit( [&]( auto& m ) { m->setResidentPrefillWeights( on != 0 ); }, r->model ); } );
}
/// fp4 policy: on != 0 (default, as in the reference) = W4A8 prefill on the fp8 matrix cores; 0 = dequantize -> bf16 GEMM
HOST_API int mila_gemma_set_fp8_activation_prefill( void* h, int on )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { std::visit( [&]( auto& m ) { m->setFp8ActivationPrefill( on != 0 ); }, r->model ); } );
}

/// mode: 0 reference-order (one launch per component), 1 fused schedule, 2 graph replay (position from device)
HOST_API int mila_gemma_decode( void* h, int32_t token, int64_t position, int mode, float* host_logits )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        upload_tokens( r, &token, 1 );
        std::visit( [&]( auto& m )
        {
            if ( mode == 0 ) m->decode( *r->tokens, position );
            else if ( mode == 1 ) m->decodeFused( *r->tokens, position );
            else
            {
                m->ensureGraph( *r->tokens, position );
                m->setDevicePosition( position );
                m->replayGraph();
            }
            m->context()->synchronize();
        }, r->model );
        download_logits( r, host_logits );
    } );
}

/// Timed decode: `warmup` untimed steps then `steps` timed ones from `start_position`, token ids
/// cycling through a fixed pattern (the sampler is outside the measured path).
/// out[0] = wall ms per step (host clock around the timed region, stream synchronised both sides)
/// out[1] = device ms per step (HIP events on the model stream)
HOST_API int mila_gemma_time_decode( void* h, int64_t start_position, int steps, int warmup, int mode, double* out )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        int32_t tok = 17;
        upload_tokens( r, &tok, 1 );
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            hipStream_t s = reinterpret_cast<hipStream_t>( ctx->getStream() );
            // every step ends with the greedy device sampler writing the next token: a real autoregressive loop
            m->setSampleInGraph( true );
            if ( mode == 2 ) m->ensureGraph( *r->tokens, start_position );
            if ( mode == 2 ) m->setDevicePosition( start_position );
            auto step = [&]( int64_t pos )
            {
                if ( mode == 0 ) { m->decode( *r->tokens, pos ); m->sampleGreedy( *r->tokens ); }
                else if ( mode == 1 ) { m->decodeFused( *r->tokens, pos ); m->sampleGreedy( *r->tokens ); }
                else { m->ensureGraph( *r->tokens, pos ); m->replayGraph(); }
            };
            int64_t pos = start_position;
            for ( int i = 0; i < warmup; ++i ) step( pos++ );
            ctx->synchronize();
            hipEvent_t e0, e1;
            hipCheck( hipEventCreate( &e0 ), "hipEventCreate" );
            hipCheck( hipEventCreate( &e1 ), "hipEventCreate" );
            const auto t0 = std::chrono::steady_clock::now();
            hipCheck( hipEventRecord( e0, s ), "hipEventRecord" );
            for ( int i = 0; i < steps; ++i ) step( pos++ );
            hipCheck( hipEventRecord( e1, s ), "hipEventRecord" );
            ctx->synchronize();
            const auto t1 = std::chrono::steady_clock::now();
            float ms = 0;
            hipCheck( hipEventElapsedTime( &ms, e0, e1 ), "hipEventElapsedTime" );
            (void)hipEventDestroy( e0 ); (void)hipEventDestroy( e1 );
            out[ 0 ] = std::chrono::duration<double, std::milli>( t1 - t0 ).count() / steps;
            out[ 1 ] = static_cast<double>( ms ) / steps;
        }, r->model );
    } );
}

/// Average launch duration of the dominant kernel (fc_gate_up fused matvec) measured with HIP events on
/// the model stream: `rounds` passes over all layers' weights (so no pass re-reads a cached matrix).
/// out[0] = average microseconds per launch, out[1] = algorithmic bytes per launch
HOST_API int mila_gemma_time_dominant_kernel( void* h, int rounds, double* out )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            hipStream_t s = reinterpret_cast<hipStream_t>( ctx->getStream() );
            const size_t L = m->layers().size();
            for ( size_t i = 0; i < L; ++i ) m->launchDominant( i );
            ctx->synchronize();
            hipEvent_t e0, e1;
            hipCheck( hipEventCreate( &e0 ), "hipEventCreate" );
            hipCheck( hipEventCreate( &e1 ), "hipEventCreate" );
            hipCheck( hipEventRecord( e0, s ), "hipEventRecord" );
            for ( int k = 0; k < rounds; ++k )
                for ( size_t i = 0; i < L; ++i ) m->launchDominant( i );
            hipCheck( hipEventRecord( e1, s ), "hipEventRecord" );
            ctx->synchronize();
            float ms = 0;
            hipCheck( hipEventElapsedTime( &ms, e0, e1 ), "hipEventElapsedTime" );
            (void)hipEventDestroy( e0 ); (void)hipEventDestroy( e1 );
            out[ 0 ] = static_cast<double>( ms ) * 1e3 / ( static_cast<double>( rounds ) * L );
            double bytes = 0;
            for ( size_t i = 0; i < L; ++i ) bytes += m->dominantBytes( i );
            out[ 1 ] = bytes / static_cast<double>( L );
        }, r->model );
    } );
}

/// a long prompt as the L6 caller feeds it (Gemma.ixx:234-267 prefill in chunks of the built prefill size, positions p0 .. p0 + n): device time of the whole
/// chunked prefill of total_T tokens from an empty cache, in ms.  The caches hold total_T positions afterwards (decode can continue at total_T).
HOST_API int mila_gemma_time_prefill_chunked( void* h, int64_t total_T, double* out_ms )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        if ( total_T <= 0 ) throw std::invalid_argument( "time_prefill_chunked: empty prompt" );
        std::vector<int32_t> toks( static_cast<size_t>( total_T ) );
        for ( int64_t i = 0; i < total_T; ++i ) toks[ i ] = static_cast<int32_t>( ( i * 7919 + 13 ) % 1000 );
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            hipStream_t s = reinterpret_cast<hipStream_t>( ctx->getStream() );
            Tensor<TensorDataType::INT32, Compute::RocmDeviceMemoryResource> dev( ctx->getDeviceId(), shape_t{ 1, total_T } );
            Compute::rocmCheck( mila_cdna4_memcpy_h2d( dev.rawData(), toks.data(), static_cast<size_t>( total_T ) * 4, ctx->getStream() ) );
            m->resetKVCache();
            ctx->synchronize();
            hipEvent_t e0, e1;
            hipCheck( hipEventCreate( &e0 ), "hipEventCreate" );
            hipCheck( hipEventCreate( &e1 ), "hipEventCreate" );
            hipCheck( hipEventRecord( e0, s ), "hipEventRecord" );
            m->prefillFrom( dev, total_T, 0 );
            hipCheck( hipEventRecord( e1, s ), "hipEventRecord" );
            ctx->synchronize();
            float ms = 0;
            hipCheck( hipEventElapsedTime( &ms, e0, e1 ), "hipEventElapsedTime" );
            (void)hipEventDestroy( e0 ); (void)hipEventDestroy( e1 );
            *out_ms = static_cast<double>( ms );
        }, r->model );
    } );
}

HOST_API int mila_gemma_time_prefill( void* h, int64_t T, int reps, double* out_ms )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        std::vector<int32_t> toks( static_cast<size_t>( T ) );
        for ( int64_t i = 0; i < T; ++i ) toks[ i ] = static_cast<int32_t>( ( i * 7919 + 13 ) % 1000 );
        upload_tokens( r, toks.data(), T );
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            hipStream_t s = reinterpret_cast<hipStream_t>( ctx->getStream() );
            m->prefill( *r->tokens, T, 0 );
            ctx->synchronize();
            hipEvent_t e0, e1;
            hipCheck( hipEventCreate( &e0 ), "hipEventCreate" );
            hipCheck( hipEventCreate( &e1 ), "hipEventCreate" );
            hipCheck( hipEventRecord( e0, s ), "hipEventRecord" );
            for ( int i = 0; i < reps; ++i ) m->prefill( *r->tokens, T, 0 );
            hipCheck( hipEventRecord( e1, s ), "hipEventRecord" );
            ctx->synchronize();
            float ms = 0;
            hipCheck( hipEventElapsedTime( &ms, e0, e1 ), "hipEventElapsedTime" );
            (void)hipEventDestroy( e0 ); (void)hipEventDestroy( e1 );
            *out_ms = static_cast<double>( ms ) / reps;
        }, r->model );
    } );
}

/// greedy generation: feeds `first_token` at `start_position`, then n_tokens - 1 more steps, each consuming the
/// token the device sampler produced; returns the sampled ids (host) -- mode as in mila_gemma_decode
HOST_API int mila_gemma_generate( void* h, int32_t first_token, int64_t start_position, int n_tokens, int mode, int32_t* host_out )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        upload_tokens( r, &first_token, 1 );
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            m->setSampleInGraph( true );
            if ( mode == 2 ) m->ensureGraph( *r->tokens, start_position );
            if ( mode == 2 ) m->setDevicePosition( start_position );
            for ( int i = 0; i < n_tokens; ++i )
            {
                const int64_t pos = start_position + i;
                if ( mode == 0 ) { m->decode( *r->tokens, pos ); m->sampleGreedy( *r->tokens ); }
                else if ( mode == 1 ) { m->decodeFused( *r->tokens, pos ); m->sampleGreedy( *r->tokens ); }
                else { m->ensureGraph( *r->tokens, pos ); m->replayGraph(); }
                Compute::rocmCheck( mila_cdna4_memcpy_d2h( host_out + i, r->tokens->data(), 4, ctx->getStream() ) );
                ctx->synchronize();
            }
        }, r->model );
    } );
}

/// Stochastic generation: like mila_gemma_generate (mode 0 or 1) with the multinomial device sampler; the per-step uniform
/// comes from a host mt19937( seed ) as in the reference's generate loop.  temperature <= 0 degenerates to greedy.
HOST_API int mila_gemma_generate_sampled( void* h, int32_t first_token, int64_t start_position, int n_tokens, int mode, float temperature, int top_k,
                                          float top_p, uint32_t seed, int32_t* host_out )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        if ( mode != 0 && mode != 1 ) throw std::invalid_argument( "generate_sampled: mode must be 0 (reference order) or 1 (fused)" );
        upload_tokens( r, &first_token, 1 );
        std::mt19937 rng( seed );
        std::uniform_real_distribution<float> uni( 0.0f, 1.0f );
        std::visit( [&]( auto& m )
        {
            auto* ctx = m->context();
            typename std::remove_reference_t<decltype( *m )>::SamplingParams sp;
            sp.temperature = temperature; sp.top_k = top_k; sp.top_p = top_p;
            for ( int i = 0; i < n_tokens; ++i )
            {
                const int64_t pos = start_position + i;
                if ( mode == 0 ) m->decode( *r->tokens, pos ); else m->decodeFused( *r->tokens, pos );
                m->sampleStochastic( *r->tokens, sp, uni( rng ) );
                Compute::rocmCheck( mila_cdna4_memcpy_d2h( host_out + i, r->tokens->data(), 4, ctx->getStream() ) );
                ctx->synchronize();
            }
        }, r->model );
    } );
}

/// model facts for the bench line: out[0] algorithmic bytes per decode token at `context`,
/// out[1] weight bytes, out[2] Linear parameter count (body), out[3] table parameter count
HOST_API int mila_gemma_info( void* h, int64_t context, double* out )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            out[ 0 ] = m->decodeBytesPerToken( context );
            out[ 1 ] = m->weightBytes();
            double p = 0;
            const auto& c = m->config();
            for ( dim_t i = 0; i < c.num_layers; ++i ) p += static_cast<double>( c.linearParamsPerLayer( c.isGlobalLayer( i ) ) );
            out[ 2 ] = p;
            out[ 3 ] = static_cast<double>( c.vocab_size ) * c.embedding_dim;
        }, r->model );
    } );
}

/// out[0..2] = getRequiredMemory(): device parameter / device state / host state bytes; out[3..5] = getMemoryStats() likewise; out[6] = the context's scratch high-water mark
HOST_API int mila_gemma_memory_stats( void* h, double* out )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            const Mila::Dnn::MemoryStats req = m->getRequiredMemory(), act = m->getMemoryStats();
            out[ 0 ] = static_cast<double>( req.device_parameter_bytes ); out[ 1 ] = static_cast<double>( req.device_state_bytes ); out[ 2 ] = static_cast<double>( req.host_state_bytes );
            out[ 3 ] = static_cast<double>( act.device_parameter_bytes ); out[ 4 ] = static_cast<double>( act.device_state_bytes ); out[ 5 ] = static_cast<double>( act.host_state_bytes );
            out[ 6 ] = static_cast<double>( m->context()->getScratchHighWaterBytes() );
        }, r->model );
    } );
}

/// nodes of the captured decode graph (0 before the first graph-mode step)
HOST_API int mila_gemma_graph_node_count( void* h, int64_t* out )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { std::visit( [&]( auto& m ) { *out = static_cast<int64_t>( m->graphNodeCount() ); }, r->model ); } );
}

/// bytes of op-owned resident prefill staging held by the layer Linears at this moment
HOST_API int mila_gemma_resident_staging_bytes( void* h, double* out )
{
    auto* r = static_cast<Runner*>( h );
    return guarded( [&] { std::visit( [&]( auto& m ) { *out = m->residentStagingBytes(); }, r->model ); } );
}

/// component names of the model, '\n'-separated, in construction order (children of each block, then temb / rmsn_final / lm_head);
/// returns the length needed (including the terminator); writes at most `cap` bytes
HOST_API int64_t mila_gemma_component_names( void* h, char* buf, int64_t cap )
{
    std::string out;
    int rc = guarded( [&]
    {
        std::visit( [&]( auto& m )
        {
            for ( auto& L : m->layers() )
            {
                out += L.getName() + "\n";
                for ( const std::string& n : L.childNames() ) out += n + "\n";
            }
            out += m->tokenEmbedding().getName() + "\n" + m->finalNorm().getName() + "\n" + m->lmHead().getName() + "\n";
        }, static_cast<Runner*>( h )->model );
    } );
    if ( rc ) return -1;
    if ( buf && cap > 0 ) { const size_t n = std::min<size_t>( out.size(), static_cast<size_t>( cap - 1 ) ); std::memcpy( buf, out.data(), n ); buf[ n ] = 0; }
    return static_cast<int64_t>( out.size() + 1 );
}

/// The leaf components used on their own, as code written against the reference uses them (construct with a name and a config,
/// setExecutionContext, build, forward): each result is compared on the host with the C-ABI launcher it resolves to, and the lifecycle
/// errors are the reference's (forward before build -> runtime_error naming the component; bad shapes -> invalid_argument).
/// Returns 0, or an error whose text says which scenario failed.
HOST_API int mila_component_scenarios( int device )
{
    return guarded( [&]
    {
        using Dev = Compute::RocmDeviceMemoryResource;
        using T16 = Tensor<TensorDataType::BF16, Dev>;
        using TI = Tensor<TensorDataType::INT32, Dev>;
        constexpr auto kD = DeviceType::Rocm;
        constexpr auto kP = TensorDataType::BF16;
        auto owned = Compute::createExecutionContext( Compute::Device::Rocm( device ) );
        auto* ctx = Compute::cast_context<kD>( owned.get() );
        const auto dev = ctx->getDeviceId();
        auto fail = [&]( const std::string& what ) { throw std::runtime_error( "component scenario failed: " + what ); };
        auto fillu = [&]( T16& t, uint64_t seed, float amp, float off ) { Compute::rocmCheck( mila_cdna4_fill_uniform_bf16( t.data(), (int64_t)t.size(), seed, amp, off, ctx->getStream() ) ); };
        auto host = [&]( const T16& t, size_t n )
        {
            std::vector<uint16_t> v( n );
            Compute::rocmCheck( mila_cdna4_memcpy_d2h( v.data(), t.rawData(), n * 2, ctx->getStream() ) );
            ctx->synchronize();
            return v;
        };
        auto expect_throw = [&]( auto&& f, bool invalid_arg, const std::string& what )
        {
            try { f(); }
            catch ( const std::invalid_argument& ) { if ( !invalid_arg ) fail( what + " (threw invalid_argument, expected runtime_error)" ); return; }
            catch ( const std::runtime_error& ) { if ( invalid_arg ) fail( what + " (threw runtime_error, expected invalid_argument)" ); return; }
            fail( what + " (did not throw)" );
        };
        const dim_t B = 2, T = 5, NH = 4, NKV = 2, HD = 64, C = NH * HD;

        // ---- Residual: forward(a, b) = a + b, component-owned output ----
        {
            Residual<kD, kP> res( "res_1", ResidualConfig{} );
            T16 a( dev, shape_t{ B, T, C } ), b( dev, shape_t{ B, T, C } ), ref( dev, shape_t{ B, T, C } );
            fillu( a, 1, 1.0f, 0.0f ); fillu( b, 2, 1.0f, 0.0f );
            res.setExecutionContext( ctx );
            expect_throw( [&] { res.forward( a, b ); }, false, "Residual::forward before build" );
            res.build( BuildContext( shape_t{ B, T, C }, RuntimeMode::Inference ) );
            auto& y = res.forward( a, b );
            Compute::rocmCheck( mila_cdna4_residual_bf16( ref.data(), a.data(), b.data(), (int64_t)a.size(), ctx->getStream() ) );
            if ( y.shape() != a.shape() || host( y, a.size() ) != host( ref, a.size() ) ) fail( "Residual::forward" );
            expect_throw( [&] { ResidualConfig().withScalingFactor( 0.0f ).validate(); }, true, "ResidualConfig scaling_factor 0" );
        }
        // ---- Swiglu<Gelu>: [.., 2H] -> [.., H] ----
        {
            Swiglu<kD, kP, ActivationType::Gelu> g( "geglu", SwigluConfig() );
            T16 x( dev, shape_t{ B, T, 2 * C } ), ref( dev, shape_t{ B, T, C } );
            fillu( x, 3, 2.0f, 0.0f );
            g.setExecutionContext( ctx );
            g.build( BuildContext( shape_t{ B, T, 2 * C }, RuntimeMode::Inference ) );
            auto& y = g.forward( x );
            Compute::rocmCheck( mila_cdna4_geglu_bf16( ref.data(), x.data(), (int)( B * T ), (int)C, ctx->getStream() ) );
            if ( y.shape() != shape_t{ B, T, C } || host( y, ref.size() ) != host( ref, ref.size() ) ) fail( "Swiglu<Gelu>::forward" );
            T16 odd( dev, shape_t{ B, T, 7 } );
            expect_throw( [&] { g.forward( odd ); }, true, "Swiglu odd width" );
        }
        // ---- Rope: prefill == decode position by position; bounds ----
        {
            const dim_t MAXS = 64;
            Rope<kD, kP> rope( "rope", RopeConfig( C, NH, NKV, MAXS ).withBase( 10000.0f ) );
            rope.setExecutionContext( ctx );
            rope.build( BuildContext( shape_t{ 1, T, C }, RuntimeMode::Inference ) );
            T16 q( dev, shape_t{ 1, T, C } ), k( dev, shape_t{ 1, T, NKV * HD } ), q1( dev, shape_t{ 1, T, C } ), k1( dev, shape_t{ 1, T, NKV * HD } );
            fillu( q, 4, 1.0f, 0.0f ); fillu( k, 5, 1.0f, 0.0f );
            Compute::rocmCheck( mila_cdna4_memcpy_d2d( q1.rawData(), q.rawData(), q.sizeInBytes(), ctx->getStream() ) );
            Compute::rocmCheck( mila_cdna4_memcpy_d2d( k1.rawData(), k.rawData(), k.sizeInBytes(), ctx->getStream() ) );
            rope.prefill( q, k, 7 );
            for ( dim_t t = 0; t < T; ++t )
            {
                auto qt = q1.slice( static_cast<size_t>( t * C ), shape_t{ 1, 1, C } );
                auto kt = k1.slice( static_cast<size_t>( t * NKV * HD ), shape_t{ 1, 1, NKV * HD } );
                rope.decode( qt, kt, 7 + t );
            }
            if ( host( q, q.size() ) != host( q1, q.size() ) || host( k, k.size() ) != host( k1, k.size() ) ) fail( "Rope::prefill vs decode" );
            expect_throw( [&] { rope.prefill( q, k, MAXS - 2 ); }, true, "Rope positions beyond the cache" );
            expect_throw( [&] { RopeConfig( C, NH, 3, MAXS ).validate(); }, true, "RopeConfig n_heads % n_kv_heads" );
        }
        // ---- GroupedQueryAttention: chunked prefill then decode == one prefill over the same tokens (same cache, same kernels' contract) ----
        {
            const dim_t MAXS = 32;
            auto mk = [&]( const char* name )
            {
                auto a = std::make_unique<GroupedQueryAttention<kD, kP>>( name, GqaConfig( C, NH, NKV ).withAttentionScale( 1.0f ) );
                a->setExecutionContext( ctx );
                a->build( BuildContext( shape_t{ 1, MAXS, ( NH + 2 * NKV ) * HD }, RuntimeMode::Inference, false, T + 1 ) );
                return a;
            };
            auto a1 = mk( "gqa" ), a2 = mk( "gqa" );
            T16 q( dev, shape_t{ 1, T + 1, C } ), k( dev, shape_t{ 1, T + 1, NKV * HD } ), v( dev, shape_t{ 1, T + 1, NKV * HD } );
            fillu( q, 6, 1.0f, 0.0f ); fillu( k, 7, 1.0f, 0.0f ); fillu( v, 8, 1.0f, 0.0f );
            auto& full = a1->prefill( q, k, v, 0 );
            auto full_h = host( full, static_cast<size_t>( ( T + 1 ) * C ) );
            a2->prefill( q.view( shape_t{ 1, T, C } ), k.view( shape_t{ 1, T, NKV * HD } ), v.view( shape_t{ 1, T, NKV * HD } ), 0 );
            auto& last = a2->decode( q.slice( static_cast<size_t>( T * C ), shape_t{ 1, 1, C } ), k.slice( static_cast<size_t>( T * NKV * HD ), shape_t{ 1, 1, NKV * HD } ),
                                     v.slice( static_cast<size_t>( T * NKV * HD ), shape_t{ 1, 1, NKV * HD } ), T );
            auto last_h = host( last, static_cast<size_t>( C ) );
            for ( dim_t i = 0; i < C; ++i )
            {
                auto f = []( uint16_t b ) { uint32_t u = (uint32_t)b << 16; float x; std::memcpy( &x, &u, 4 ); return x; };
                const float d = std::fabs( f( last_h[ i ] ) - f( full_h[ static_cast<size_t>( T * C + i ) ] ) );
                if ( !( d <= 2e-2f ) ) fail( "GroupedQueryAttention decode vs prefill row (|d| = " + std::to_string( d ) + ")" );
            }
            if ( a2->cacheLength() != T + 1 || !a2->supportsKVCache() ) fail( "GroupedQueryAttention cache bookkeeping" );
            if ( !a2->rewindKvCache( 2 ) || a2->cacheLength() != 2 || a2->rewindKvCache( 9 ) ) fail( "GroupedQueryAttention::rewindKvCache" );
            a2->resetKVCache();
            if ( a2->cacheLength() != 0 ) fail( "GroupedQueryAttention::resetKVCache" );
            expect_throw( [&] { GqaConfig( C, NH, 3 ).validate(); }, true, "GqaConfig num_heads % num_kv_heads" );
            GroupedQueryAttention<kD, kP> unbuilt( "gqa", GqaConfig( C, NH, NKV ) );
            expect_throw( [&] { unbuilt.prefill( q, k, v, 0 ); }, false, "GroupedQueryAttention::prefill before build" );
        }
        // ---- TokenEmbedding: forward(tokens) = wte[id] * scale; bf16 and FP8 tables; the tied Linear adopts the table ----
        {
            const dim_t V = 50, E = 32;
            std::vector<uint16_t> table( static_cast<size_t>( V * E ) );
            for ( size_t i = 0; i < table.size(); ++i ) { const float x = 0.01f * static_cast<float>( ( i * 37 ) % 101 ) - 0.5f; uint32_t u; std::memcpy( &u, &x, 4 ); table[ i ] = static_cast<uint16_t>( u >> 16 ); }
            std::vector<int32_t> ids{ 3, 49, 0, 17, 17, 8 };
            TI tok( dev, shape_t{ 2, 3 } );
            Compute::rocmCheck( mila_cdna4_memcpy_h2d( tok.rawData(), ids.data(), ids.size() * 4, ctx->getStream() ) );
            TokenEmbedding<kD, TensorDataType::INT32, kP> emb( "temb", TokenEmbeddingConfig().withVocabSize( V ).withEmbeddingDim( E ).withEmbeddingScale( 2.0f ) );
            emb.setExecutionContext( ctx );
            expect_throw( [&] { emb.forward( tok ); }, false, "TokenEmbedding::forward before build" );
            emb.build( BuildContext( shape_t{ 2, 4 }, RuntimeMode::Inference ) );
            emb.loadParameter( "wte", table.data(), table.size() * 2 );
            auto& y = emb.forward( tok );
            auto yh = host( y, ids.size() * static_cast<size_t>( E ) );
            auto f = []( uint16_t b ) { uint32_t u = (uint32_t)b << 16; float x; std::memcpy( &x, &u, 4 ); return x; };
            for ( size_t t = 0; t < ids.size(); ++t )
                for ( dim_t e = 0; e < E; ++e )
                    if ( f( yh[ t * E + e ] ) != 2.0f * f( table[ static_cast<size_t>( ids[ t ] * E + e ) ] ) ) fail( "TokenEmbedding::forward (bf16 table)" );
            if ( emb.takeError() != 0 ) fail( "TokenEmbedding error flag set on valid ids" );
            TI big( dev, shape_t{ 2, 5 } );
            expect_throw( [&] { emb.forward( big ); }, false, "TokenEmbedding input beyond the built shape" );
            expect_throw( [&] { emb.loadParameter( "wte_scale", table.data(), 4 ); }, true, "wte_scale on an unquantized table" );
            // tied head: logits = x . wte^T through the SAME allocation
            Linear<kD, kP> head( "lm_head", LinearConfig( E, V ).withBias( false ) );
            head.setExecutionContext( ctx );
            head.installSharedWeight( emb.getWeightTensorShared() );
            head.build( BuildContext( shape_t{ 1, 1, E }, RuntimeMode::Inference ) );
            if ( head.getWeight().rawData() != emb.getWeightTensorShared()->rawData() || !head.hasSharedWeight() ) fail( "Linear::installSharedWeight does not alias the table" );
            // FP8 table: quantize on load, gather dequantizes with the row scale
            TokenEmbedding<kD, TensorDataType::INT32, kP, PerChannelFp8<>> q8( "temb", TokenEmbeddingConfig().withVocabSize( V ).withEmbeddingDim( E ) );
            q8.setExecutionContext( ctx );
            q8.build( BuildContext( shape_t{ 2, 4 }, RuntimeMode::Inference ) );
            q8.loadParameter( "wte", table.data(), table.size() * 2 );
            auto& y8 = q8.forward( tok );
            auto y8h = host( y8, ids.size() * static_cast<size_t>( E ) );
            for ( size_t t = 0; t < ids.size(); ++t )
                for ( dim_t e = 0; e < E; ++e )
                {
                    const float want = f( table[ static_cast<size_t>( ids[ t ] * E + e ) ] );
                    if ( std::fabs( f( y8h[ t * E + e ] ) - want ) > 0.0625f * 0.5f + 1e-3f ) fail( "TokenEmbedding::forward (FP8 table)" );   // e4m3: 3 mantissa bits, |w| <= 0.5
                }
            Linear<kD, kP, PerChannelFp8<>> head8( "lm_head", LinearConfig( E, V ).withBias( false ) );
            head8.setExecutionContext( ctx );
            head8.installSharedWeight( q8.getWeightTensorShared(), q8.getWeightScalesTensorShared() );
            head8.build( BuildContext( shape_t{ 1, 1, E }, RuntimeMode::Inference ) );
            if ( head8.getWeightScale()->rawData() != q8.getWeightScalesTensorShared()->rawData() ) fail( "Linear::installSharedWeight(weight, scales)" );
            expect_throw( [&] { head8.installSharedWeight( q8.getWeightTensorShared(), q8.getWeightScalesTensorShared() ); }, false, "installSharedWeight after build" );
        }
        // ---- GemmaBlock<kGlobal> as its own type: Tests/Dnn/Components/Transformers/Gemma/Gemma.Block.Cuda.cpp:132-262 on the same small geometry ----
        {
            using LocalBlock = GemmaBlock<kD, kP, false>;
            using GlobalBlock = GemmaBlock<kD, kP, true>;
            static_assert( LocalBlock::getDeviceType() == kD && LocalBlock::getPrecision() == kP );
            auto cfg_of = []( bool g )
            {
                GemmaBlockConfig c;
                c.model_dim = 64; c.hidden_dim = 128; c.num_heads = 4; c.num_kv_heads = g ? 1 : 2; c.head_dim = g ? 64 : 32;
                c.window = g ? 0 : 8; c.rotary_dim = g ? 32 : 0; c.rope_theta = g ? 1000000.0f : 10000.0f; c.rms_norm_eps = 1e-6f; c.max_seq = 32;
                return c;
            };
            const dim_t seq = 8;
            LocalBlock local( "gemma_local", cfg_of( false ) );
            GlobalBlock global( "gemma_global", cfg_of( true ) );
            local.setExecutionContext( ctx ); global.setExecutionContext( ctx );
            if ( local.isBuilt() || global.isBuilt() ) fail( "GemmaBlock built before build()" );                                       // ConstructLocal / ConstructGlobal
            {
                LocalBlock rank2( "gemma_local", cfg_of( false ) ); rank2.setExecutionContext( ctx );
                expect_throw( [&] { rank2.build( BuildContext( shape_t{ seq, 64 }, RuntimeMode::Inference ) ); }, true, "GemmaBlock Build_ThrowsOnNonRank3Input (:185)" );
                LocalBlock wrong( "gemma_local", cfg_of( false ) ); wrong.setExecutionContext( ctx );
                expect_throw( [&] { wrong.build( BuildContext( shape_t{ 1, seq, 65 }, RuntimeMode::Inference ) ); }, true, "GemmaBlock Build_ThrowsOnModelDimMismatch (:194)" );
            }
            // LocalGeometry_SlidingWidthsAndWindow (:206), GlobalGeometry_WidenedHeadDimAndKEqualsV (:220), HeadDim_IsDecoupledFromResidualStream (:235)
            if ( local.isGlobal() || local.headDim() != 32 || local.numKVHeads() != 2 || local.keyEqualsValue() || local.window() != 8 || local.qProjWidth() != 128 ||
                 local.kvProjWidth() != 64 || local.packedQKVWidth() != 256 ) fail( "GemmaBlock local geometry" );
            if ( !global.isGlobal() || global.headDim() != 64 || global.numKVHeads() != 1 || !global.keyEqualsValue() || global.window() != 0 || global.qProjWidth() != 256 ||
                 global.kvProjWidth() != 64 || global.packedQKVWidth() != 320 ) fail( "GemmaBlock global geometry (K = V drops the V section)" );
            if ( local.headDim() == 64 / 4 || local.qProjWidth() == 64 ) fail( "GemmaBlock head_dim must be decoupled from model_dim / num_heads" );
            // GetComponents_ReturnsCorrectChildrenSize (:249), GlobalBlock_HasSameGraphShape (:258): 7 norms + qkv_proj + rope + gqa + o_proj + res_1 + fc_gate_up + geglu + fc_down + res_2
            if ( local.childNames().size() != 16u || global.childNames().size() != 16u ) fail( "GemmaBlock children" );
            local.build( BuildContext( shape_t{ 1, seq, 64 }, RuntimeMode::Inference ) );                                                 // BuildLocal_SetsIsBuilt / AllocatesParameters
            global.build( BuildContext( shape_t{ 1, seq, 64 }, RuntimeMode::Inference ) );
            if ( !local.isBuilt() || !global.isBuilt() || local.qkv_proj->getParameterBytes() != 256u * 64u * 2u || global.qkv_proj->getParameterBytes() != 320u * 64u * 2u )
                fail( "GemmaBlock build / parameter allocation" );
            // SaveThenLoad_RestoresLayerScalar (:357): the scalar travels as a one-element F32 parameter
            const float two_and_a_half = 2.5f;
            local.loadParameter( "layer_scalar", &two_and_a_half, 4 );
            if ( local.layer_scalar != 2.5f ) fail( "GemmaBlock layer_scalar" );
            expect_throw( [&] { local.loadParameter( "layer_scalar", &two_and_a_half, 8 ); }, true, "GemmaBlock layer_scalar blob size" );
            expect_throw( [&] { local.loadParameter( "no_such", &two_and_a_half, 4 ); }, true, "GemmaBlock unknown parameter" );
            // a forward through both kinds: prefill of the chunk, then one decode step continues it (finite outputs of the right shape)
            auto fill_w = [&]( auto& blk, uint64_t seed )
            {
                for ( auto* lin : { blk.qkv_proj.get(), blk.o_proj.get(), blk.fc_gate_up.get(), blk.fc_down.get() } ) fillu( lin->getWeight(), seed++, 0.1f, 0.0f );
                for ( auto* n : { blk.input_norm.get(), blk.q_norm.get(), blk.k_norm.get(), blk.v_norm.get(), blk.post_attn_norm.get(), blk.pre_ffn_norm.get(), blk.post_ffn_norm.get() } )
                    fillu( *n->getWeight(), seed++, 0.1f, 1.0f );
            };
            // (the attention kernels serve head sizes 64 ... 512: the forward runs on the same graph with head_dim 64 / 128)
            auto fwd_cfg = [&]( bool g ) { auto c = cfg_of( g ); c.head_dim = g ? 128 : 64; return c; };
            LocalBlock flocal( "gemma_local", fwd_cfg( false ) );
            GlobalBlock fglobal( "gemma_global", fwd_cfg( true ) );
            flocal.setExecutionContext( ctx ); fglobal.setExecutionContext( ctx );
            flocal.build( BuildContext( shape_t{ 1, seq, 64 }, RuntimeMode::Inference ) );
            fglobal.build( BuildContext( shape_t{ 1, seq, 64 }, RuntimeMode::Inference ) );
            fill_w( flocal, 100 ); fill_w( fglobal, 200 );
            T16 x( dev, shape_t{ 1, seq, 64 } ), x1( dev, shape_t{ 1, 1, 64 } );
            fillu( x, 7, 1.0f, 0.0f ); fillu( x1, 8, 1.0f, 0.0f );
            for ( IDecoderLayer<kD, kP>* blk : { static_cast<IDecoderLayer<kD, kP>*>( &flocal ), static_cast<IDecoderLayer<kD, kP>*>( &fglobal ) } )
            {
                auto& y = blk->prefill( x, 0 );
                if ( y.shape() != shape_t{ 1, seq, 64 } ) fail( "GemmaBlock::prefill output shape" );
                auto yh = host( y, static_cast<size_t>( seq * 64 ) );
                auto& d = blk->decode( x1, seq );
                if ( d.shape() != shape_t{ 1, 1, 64 } ) fail( "GemmaBlock::decode output shape" );
                auto dh = host( d, 64 );
                for ( uint16_t b : yh ) if ( ( b & 0x7f80 ) == 0x7f80 ) fail( "GemmaBlock::prefill produced a non-finite value" );
                for ( uint16_t b : dh ) if ( ( b & 0x7f80 ) == 0x7f80 ) fail( "GemmaBlock::decode produced a non-finite value" );
                blk->resetKVCache();
            }
            expect_throw( [&] { flocal.decode( x, 0 ); }, true, "GemmaBlock::decode takes one token" );
        }
        ctx->synchronize();
    } );
}

// ---- L6: GemmaModel (Models/GemmaModel.ixx): fromPretrained / generate -------------------------------------------------------------
using RocmGemmaModel = GemmaModel<DeviceType::Rocm, TensorDataType::BF16>;

static GemmaModelConfig model_config_of( int policy, int64_t context, int64_t chunk, int bounded )
{
    if ( policy < 0 || policy > 2 ) throw std::invalid_argument( "unknown weight policy" );
    GemmaModelConfig mc;
    mc.withContextLength( context ).withWeightQuantization( policy == 0 ? WeightQuantization::None : ( policy == 1 ? WeightQuantization::FP8 : WeightQuantization::FP4 ) );
    mc.withPrefillChunk( chunk ).withBoundedLocalKv( bounded != 0 );
    return mc;
}

/// GemmaModel::fromPretrained( path, GemmaModelConfig( context ).withWeightQuantization( policy ) ): geometry from the artifact's metadata
HOST_API void* mila_gemma_model_from_pretrained( const char* path, int policy, int64_t context, int64_t prefill_chunk, int bounded_local_kv, int device )
{
    RocmGemmaModel* m = nullptr;
    int rc = guarded( [&] { m = RocmGemmaModel::fromPretrained( path, model_config_of( policy, context, prefill_chunk, bounded_local_kv ), Compute::Device::Rocm( device ) ).release(); } );
    return rc == 0 ? m : nullptr;
}

/// the same model over synthetic parameters; profile as in mila_gemma_init_synthetic (NULL = unit profile)
HOST_API void* mila_gemma_model_synthetic( int policy, const mila_gemma_config* c, int64_t context, int64_t prefill_chunk, uint64_t seed, const float* p, int device )
{
    RocmGemmaModel* m = nullptr;
    int rc = guarded( [&]
    {
        GemmaConfig cfg;
        int bounded = 0;
        if ( c )
        {
            cfg.vocab_size = c->vocab_size; cfg.embedding_dim = c->embedding_dim; cfg.num_layers = c->num_layers; cfg.num_heads = c->num_heads;
            cfg.num_kv_heads = c->num_kv_heads; cfg.head_dim = c->head_dim; cfg.hidden_dim = c->hidden_dim; cfg.global_head_dim = c->global_head_dim;
            cfg.num_global_kv_heads = c->num_global_kv_heads; cfg.window = c->window; cfg.sliding_window_pattern = c->sliding_window_pattern;
            cfg.global_rotary_dim = c->global_rotary_dim;
            bounded = c->bounded_local_kv != 0;
        }
        GemmaTransformer<NoWeightQuant>::SyntheticProfile pr;
        if ( p ) { pr.linear_gain = p[ 0 ]; pr.qk_norm_center = p[ 1 ]; pr.post_norm_center = p[ 2 ]; pr.layer_scalar = p[ 3 ]; pr.table_gain = p[ 4 ]; }
        m = RocmGemmaModel::fromSynthetic( cfg, model_config_of( policy, context, prefill_chunk, bounded ), seed, pr, Compute::Device::Rocm( device ) ).release();
    } );
    return rc == 0 ? m : nullptr;
}

HOST_API void mila_gemma_model_destroy( void* h ) { delete static_cast<RocmGemmaModel*>( h ); }

/// generate(): max_new < 0 = no budget (run to a stop token or the context bound); n_stop == 0 = the model's default stop set; the callback's tokens
/// are appended to out_tokens (at most cap).  *out_status = GenerateStatus; *out_reused = prompt tokens served from the KV caches; cancel_after >= 0: the
/// client's stop request is raised from inside on_token once that many tokens were delivered (the std::stop_token of the reference's generate)
HOST_API int mila_gemma_model_generate( void* h, const int32_t* prompt, int64_t n_prompt, int max_new, const int32_t* stop_tokens, int n_stop, float temperature, int top_k,
                                        float top_p, int64_t seed, int32_t* out_tokens, int64_t cap, int64_t* out_count, int32_t* out_status, int64_t* out_reused,
                                        int64_t cancel_after )
{
    return guarded( [&]
    {
        auto* m = static_cast<RocmGemmaModel*>( h );
        if ( !m || !prompt || !out_count || !out_status ) throw std::invalid_argument( "gemma_model_generate: null argument" );
        GenerateParams gp;
        if ( max_new >= 0 ) gp.max_new_tokens = max_new;
        gp.sampling.temperature = temperature; gp.sampling.top_k = top_k; gp.sampling.top_p = top_p;
        for ( int i = 0; i < n_stop; ++i ) gp.stop_tokens.push_back( stop_tokens[ i ] );
        if ( seed >= 0 ) m->seedSampler( static_cast<uint64_t>( seed ) );
        int64_t n = 0;
        std::atomic<bool> stop{ cancel_after == 0 };
        const GenerateStatus st = m->generate( std::span<const int32_t>( prompt, static_cast<size_t>( n_prompt ) ),
                                               [&]( int32_t t ) { if ( out_tokens && n < cap ) out_tokens[ n ] = t; ++n; if ( cancel_after >= 0 && n >= cancel_after ) stop.store( true ); }, gp, &stop );
        *out_count = n;
        *out_status = static_cast<int32_t>( st );
        if ( out_reused ) *out_reused = m->lastReusedPrefix();
    } );
}

}  // extern "C"
