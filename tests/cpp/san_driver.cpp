// AddressSanitizer + UndefinedBehaviorSanitizer driver for the CPU-only pieces: the container readers /
// writers of Mila/Serialization.h (a hand-written JSON parser and a binary index parser over an mmap) and the C oracle.
// Built by tests/test_sanitizers_cpu.py with -fsanitize=address,undefined -fno-sanitize-recover=all (CPU container only: GPU
// sanitizers are not available on the pool).  TEST INFRASTRUCTURE.
//   san_driver list  <file>            parse either container, touch every byte of every tensor
//   san_driver copy  <src> <dst>       safetensors reader -> writer
//   san_driver tobin <src> <dst>       either container -> MILA .bin
//   san_driver oracle                  a battery of oracle calls at small, ragged and degenerate sizes
// exit 0 = accepted, 3 = rejected with an exception (the correct answer for a malformed file); a sanitizer report aborts.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "Mila/Serialization.h"
#include "mila_oracle.h"

using namespace Mila::Dnn::Serialization;

static int run_oracle()
{
    // ragged and degenerate shapes on purpose: sizes of 1, odd sizes, window larger than the history, ring wrap
    unsigned sink = 0;
    for ( int K : { 8, 24, 128 } )
        for ( int N : { 1, 3, 9 } )
            for ( int M : { 1, 5, 8 } )
            {
                std::vector<float> X( (size_t)M * K ), W( (size_t)N * K ), B( N ), Y( (size_t)M * N );
                for ( size_t i = 0; i < X.size(); ++i ) X[ i ] = std::sin( 0.3f * i );
                for ( size_t i = 0; i < W.size(); ++i ) W[ i ] = std::cos( 0.7f * i ) * 0.1f;
                for ( int i = 0; i < N; ++i ) B[ i ] = 0.01f * i;
                orc_cpu_linear( Y.data(), X.data(), W.data(), B.data(), M, K, N );
                orc_cpu_linear_naive( Y.data(), X.data(), W.data(), nullptr, M, K, N );
                std::vector<uint16_t> Wb( W.size() );
                orc_f32_to_bf16_array( Wb.data(), W.data(), (int64_t)W.size() );
                orc_linear_bf16w( Y.data(), X.data(), Wb.data(), nullptr, M, K, N );
                std::vector<uint8_t> q8( W.size() ), q4( W.size() / 2 );
                std::vector<float> s8( N ), s4( (size_t)N * ( K / 8 ) );
                orc_quantize_fp8_per_channel( q8.data(), s8.data(), Wb.data(), N, K );
                orc_linear_fp8w( Y.data(), X.data(), q8.data(), s8.data(), nullptr, M, K, N );
                if ( K % 64 == 0 )
                {
                    s4.assign( (size_t)N * ( K / 64 ), 0.0f );
                    orc_quantize_fp4_per_group( q4.data(), s4.data(), Wb.data(), N, K, 64 );
                    orc_linear_fp4w( Y.data(), X.data(), q4.data(), s4.data(), nullptr, M, K, N, 64 );
                    std::vector<uint8_t> w8( W.size() );
                    orc_upcast_fp4_to_fp8( w8.data(), q4.data(), s4.data(), orc_fp8_weight_scale_from_groups( s4.data(), (int64_t)s4.size() ), N, K, 64 );
                    std::vector<uint8_t> x8( X.size() ); std::vector<float> ts( M );
                    orc_quantize_act_fp8_per_token( x8.data(), ts.data(), X.data(), M, K );
                    orc_linear_fp8a_fp8w( Y.data(), x8.data(), ts.data(), w8.data(), nullptr, 1.0f, nullptr, M, K, N );
                }
                sink += (unsigned)Y[ 0 ];
            }
    for ( int dim : { 1, 7, 64 } )
        for ( int inner : { 1, 3 } )
        {
            const int outer = 3;
            std::vector<float> X( (size_t)outer * dim * inner ), Y( X.size() ), w( dim, 1.0f ), b( dim, 0.0f ), m( outer * inner ), r( outer * inner );
            for ( size_t i = 0; i < X.size(); ++i ) X[ i ] = std::sin( 0.11f * i ) * 3.0f;
            orc_cpu_softmax( Y.data(), X.data(), outer, dim, inner );
            orc_cpu_layernorm( Y.data(), m.data(), r.data(), X.data(), w.data(), b.data(), outer, dim, inner, 1e-5f );
            orc_rmsnorm( Y.data(), r.data(), X.data(), w.data(), nullptr, outer, dim, inner, 1e-6f, 0.0f );
            orc_cpu_gelu( Y.data(), X.data(), (int64_t)X.size() );
            orc_cpu_residual( Y.data(), X.data(), X.data(), (int64_t)X.size() );
        }
    {
        const int B = 2, T = 5, C = 8, NH = 2;
        std::vector<float> X( (size_t)B * T * 3 * C ), Y( (size_t)B * T * C );
        for ( size_t i = 0; i < X.size(); ++i ) X[ i ] = std::sin( 0.2f * i );
        orc_cpu_mha( Y.data(), X.data(), B, T, C, NH );
    }
    for ( int window : { 0, 3, 100 } )
    {
        const int B = 1, Tq = 4, Tk = 9, NH = 4, NKV = 2, HS = 8, cap = 5;
        std::vector<float> q( (size_t)B * Tq * NH * HS ), k( (size_t)B * Tk * NKV * HS ), v( k.size() ), out( (size_t)B * Tq * NH * HS );
        for ( size_t i = 0; i < q.size(); ++i ) q[ i ] = std::sin( 0.3f * i );
        for ( size_t i = 0; i < k.size(); ++i ) { k[ i ] = std::cos( 0.2f * i ); v[ i ] = std::sin( 0.5f * i ); }
        orc_gqa_attention( out.data(), q.data(), k.data(), v.data(), B, Tq, Tk, NH, NKV, HS, Tk - Tq, window, 0.35f );
        std::vector<float> Kc( (size_t)B * NKV * cap * HS, 0.0f ), Vc( Kc.size(), 0.0f ), lin( (size_t)B * cap * NKV * HS );
        for ( int s = 0; s < Tk; s += 4 ) orc_kv_write( Kc.data(), Vc.data(), k.data() + (size_t)s * NKV * HS, v.data() + (size_t)s * NKV * HS, B, std::min( 4, Tk - s ), NKV, HS, s, cap );
        orc_kv_ring_to_linear( lin.data(), Kc.data(), B, NKV, HS, cap, Tk - cap, cap );
    }
    {
        const int HS = 16, max_seq = 12;
        std::vector<float> c( (size_t)max_seq * HS / 2 ), s( c.size() ), X( (size_t)2 * 3 * 2 * HS ), Y( X.size() );
        orc_rope_build_cache( c.data(), s.data(), max_seq, HS, 10000.0f, 8 );
        for ( size_t i = 0; i < X.size(); ++i ) X[ i ] = std::sin( 0.3f * i );
        orc_rope_rotate( Y.data(), X.data(), c.data(), s.data(), 2, 3, 2, HS, 9 );
    }
    {
        std::vector<float> lg( 301 ); double mg[ 3 ];
        for ( size_t i = 0; i < lg.size(); ++i ) lg[ i ] = std::sin( 0.37f * i ) * 5.0f;
        for ( float r : { 0.0f, 0.5f, 0.999999f } ) sink += (unsigned)orc_sample_stochastic( lg.data(), (int)lg.size(), 30.0f, 0.8f, 7, 0.9f, r, mg );
        int32_t tok[ 3 ] = { 0, 300, 5 }; std::vector<float> tb( 301 * 4, 0.5f ), Y( 12 );
        sink += (unsigned)orc_embedding_gather( Y.data(), tok, tb.data(), 3, 4, 301, 2.0f );
        tok[ 1 ] = 301;
        sink += (unsigned)orc_embedding_gather( Y.data(), tok, tb.data(), 3, 4, 301, 0.0f );      // out-of-range id: error code, no access
    }
    for ( uint32_t b = 0; b < 65536; b += 7 ) sink += orc_f32_to_e4m3( orc_bf16_to_f32( (uint16_t)b ) ) + orc_f32_to_e2m1( orc_bf16_to_f32( (uint16_t)b ) );
    std::printf( "oracle battery done (%u)\n", sink );
    return 0;
}

int main( int argc, char** argv )
{
    if ( argc < 2 ) return 2;
    const std::string cmd = argv[ 1 ];
    try
    {
        if ( cmd == "oracle" ) return run_oracle();
        if ( cmd == "list" && argc == 3 )
        {
            PretrainedModelReader r( argv[ 2 ] );
            unsigned long sum = 0;
            for ( auto& e : r.entries() )
            {
                const auto* p = static_cast<const unsigned char*>( e.data );
                for ( size_t i = 0; i < e.nbytes(); ++i ) sum += p[ i ];                      // every mapped byte a loader would read
                (void)e.elements();
            }
            (void)toMetadataJSON( r.getPretrainedMetadata() );
            std::printf( "accepted: %zu tensors, checksum %lu, container %s\n", r.entries().size(), sum, r.isMilaContainer() ? "mila" : "safetensors" );
            return 0;
        }
        if ( cmd == "copy" && argc == 4 )
        {
            SafeTensorsReader rd( argv[ 2 ] );
            SafeTensorsWriter wr( argv[ 3 ] );
            for ( auto& e : rd.entries() ) wr.declareTensor( e.name, e.dtype, e.shape );
            for ( auto& [ k, v ] : rd.metadata() ) wr.setMetadata( k, v );
            wr.beginData();
            for ( auto& e : rd.entries() ) wr.writeTensorData( e.name, e.data, e.nbytes() );
            wr.close();
            return 0;
        }
        if ( cmd == "tobin" && argc == 4 )
        {
            PretrainedModelReader rd( argv[ 2 ] );
            MilaBinWriter wr( argv[ 3 ] );
            for ( auto& e : rd.entries() ) wr.declareTensor( e.name, e.dtype, e.shape );
            wr.setMetadataJSON( rd.metadataJSON().empty() ? std::string( "{}" ) : rd.metadataJSON() );
            wr.beginData();
            for ( auto& e : rd.entries() ) wr.writeTensorData( e.name, e.data, e.nbytes() );
            wr.close();
            return 0;
        }
        if ( cmd == "metadata" && argc == 3 )
        {
            std::FILE* f = std::fopen( argv[ 2 ], "rb" );
            if ( !f ) return 3;
            std::string t; char buf[ 4096 ]; size_t n;
            while ( ( n = std::fread( buf, 1, sizeof buf, f ) ) > 0 ) t.append( buf, n );
            std::fclose( f );
            std::printf( "%s\n", toMetadataJSON( parseMetadataJSON( t ) ).c_str() );
            return 0;
        }
    }
    catch ( const std::exception& e )
    {
        std::printf( "rejected: %s\n", e.what() );
        return 3;
    }
    return 2;
}
