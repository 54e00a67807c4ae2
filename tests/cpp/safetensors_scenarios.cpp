// The 16 container scenarios of the reference's Tests/Dnn/Serialization/SafeTensors.Cpu.cpp (:133-532), one function each under the
// reference's test name, run against the host mirror's containers (mila_amd/host/include/Mila/Serialization.h: SafeTensorsWriter,
// PretrainedModelReader, toMetadataJSON).  Same call sequences, same values, same expectations -- including the exception TYPE the
// reference's tests expect (std::runtime_error).  The mirror names dtypes by their safetensors strings ("F32", "U8", "I32") where the
// reference passes TensorDataType; a blob is the reader's mapped bytes.  The legacy MILA file is written byte by byte from the format
// definition, as in the reference's own test (:81-127).  TEST INFRASTRUCTURE (built and driven by tests/test_serialization_cpu.py).
//   safetensors_scenarios <scenario name> <scratch dir>      exit 0 = passed, 1 = an expectation failed (printed), 2 = unknown name
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include <unistd.h>

#include "Mila/Serialization.h"

using namespace Mila::Dnn::Serialization;

static int g_failures = 0;
#define EXPECT( cond ) do { if ( !( cond ) ) { std::printf( "  EXPECT failed at line %d: %s\n", __LINE__, #cond ); ++g_failures; } } while ( 0 )
#define EXPECT_THROW_RUNTIME( stmt ) do { bool thrown_ = false; try { stmt; } catch ( const std::runtime_error& ) { thrown_ = true; } catch ( ... ) { std::printf( "  line %d: threw something other than std::runtime_error\n", __LINE__ ); } \
    if ( !thrown_ ) { std::printf( "  EXPECT_THROW(std::runtime_error) failed at line %d: %s\n", __LINE__, #stmt ); ++g_failures; } } while ( 0 )
static bool floatEq( float a, float b ) { return std::fabs( a - b ) <= 4.0f * 1.1920929e-7f * std::fmax( std::fabs( a ), std::fabs( b ) ); }     // EXPECT_FLOAT_EQ: 4 ulp

static std::string g_dir;
static std::string scratch( const std::string& stem ) { return g_dir + "/mila_safetensors_" + stem + ".bin"; }

// SafeTensors.Cpu.cpp:81-127 writeMilaFormatFile
static void writeMilaFormatFile( const std::string& path, const std::string& metadata_json, const std::string& tensor_name, const std::vector<float>& values )
{
    std::FILE* f = std::fopen( path.c_str(), "wb" );
    if ( !f ) throw std::runtime_error( "cannot create " + path );
    const uint32_t magic = 0x4D494C41, version = 1, num_tensors = 1;
    std::fwrite( &magic, 4, 1, f ); std::fwrite( &version, 4, 1, f ); std::fwrite( &num_tensors, 4, 1, f );
    const uint32_t metadata_size = static_cast<uint32_t>( metadata_json.size() );
    std::fwrite( &metadata_size, 4, 1, f ); std::fwrite( metadata_json.data(), 1, metadata_json.size(), f );
    const uint32_t name_length = static_cast<uint32_t>( tensor_name.size() ), dtype_code = 0 /* float32 */, ndim = 2, dim0 = 2, dim1 = static_cast<uint32_t>( values.size() / 2 );
    const long index_start = std::ftell( f );
    const uint64_t index_size = 4 + name_length + 4 + 4 + 4 * ndim + 8 + 8;
    const uint64_t data_offset = static_cast<uint64_t>( index_start ) + index_size, nbytes = values.size() * sizeof( float );
    std::fwrite( &name_length, 4, 1, f ); std::fwrite( tensor_name.data(), 1, tensor_name.size(), f );
    std::fwrite( &dtype_code, 4, 1, f ); std::fwrite( &ndim, 4, 1, f ); std::fwrite( &dim0, 4, 1, f ); std::fwrite( &dim1, 4, 1, f );
    std::fwrite( &data_offset, 8, 1, f ); std::fwrite( &nbytes, 8, 1, f );
    EXPECT( static_cast<uint64_t>( std::ftell( f ) ) == data_offset );
    std::fwrite( values.data(), 1, static_cast<size_t>( nbytes ), f );
    std::fclose( f );
}

// :133
static void RoundTripsTensorsOfMixedDataTypes()
{
    const std::string path = scratch( "mixed" );
    const std::vector<float> weights{ 1.0f, -2.5f, 3.25f, 4.75f, -5.5f, 6.0f };
    const std::vector<uint8_t> packed{ 0x0F, 0xA3, 0x71, 0xC2 };
    const std::vector<int32_t> counts{ 7, -11, 13 };
    {
        SafeTensorsWriter writer( path );
        writer.declareTensor( "block.weight", "F32", { 2, 3 } );
        writer.declareTensor( "block.weight.packed", "U8", { 4 } );
        writer.declareTensor( "block.counts", "I32", { 3 } );
        EXPECT( writer.getTensorCount() == 3u );
        writer.beginData();
        writer.writeTensorData( "block.weight", weights.data(), weights.size() * sizeof( float ) );
        writer.writeTensorData( "block.weight.packed", packed.data(), packed.size() );
        writer.writeTensorData( "block.counts", counts.data(), counts.size() * sizeof( int32_t ) );
        writer.close();
    }
    PretrainedModelReader reader( path );
    EXPECT( reader.hasTensor( "block.weight" ) ); EXPECT( reader.hasTensor( "block.weight.packed" ) ); EXPECT( reader.hasTensor( "block.counts" ) );
    EXPECT( reader.getTensorNames().size() == 3u );
    const auto& w = reader.get( "block.weight" );
    EXPECT( w.dtype == "F32" ); EXPECT( w.shape.size() == 2 && w.shape[ 0 ] == 2 && w.shape[ 1 ] == 3 );
    EXPECT( w.nbytes() == weights.size() * sizeof( float ) ); EXPECT( std::memcmp( w.data, weights.data(), w.nbytes() ) == 0 );
    const auto& p = reader.get( "block.weight.packed" );
    EXPECT( p.dtype == "U8" ); EXPECT( p.nbytes() == packed.size() ); EXPECT( std::memcmp( p.data, packed.data(), p.nbytes() ) == 0 );
    const auto& c = reader.get( "block.counts" );
    EXPECT( c.dtype == "I32" ); EXPECT( c.nbytes() == counts.size() * sizeof( int32_t ) ); EXPECT( std::memcmp( c.data, counts.data(), c.nbytes() ) == 0 );
}

static void writeOneFp32( const std::string& path, const std::vector<float>& values, const char* key = nullptr, const std::string& value = "" )
{
    SafeTensorsWriter writer( path );
    writer.declareTensor( "w", "F32", { static_cast<int64_t>( values.size() ) } );
    if ( key ) writer.setMetadata( key, value );
    writer.beginData();
    writer.writeTensorData( "w", values.data(), values.size() * sizeof( float ) );
    writer.close();
}

// :187
static void CarriesMilaConfigThroughMetadata()
{
    const std::string path = scratch( "config" );
    const std::string config = R"({"architecture":"llama","model_name":"tiny","vocab_size":128,)"
                               R"("embedding_dim":64,"num_layers":2,"num_heads":4,"num_kv_heads":2,)"
                               R"("rope_theta":10000.0,"tie_word_embeddings":true})";
    writeOneFp32( path, { 1.0f, 2.0f }, kMilaConfigMetadataKey, config );
    PretrainedModelReader reader( path );
    const auto& m = reader.getPretrainedMetadata();
    EXPECT( m.architecture == "llama" ); EXPECT( m.model_name == "tiny" ); EXPECT( m.vocab_size == 128u ); EXPECT( m.embedding_dim == 64u );
    EXPECT( m.num_layers == 2u ); EXPECT( m.num_heads == 4u ); EXPECT( m.num_kv_heads == 2u ); EXPECT( m.tie_word_embeddings ); EXPECT( floatEq( m.rope_theta, 10000.0f ) );
}

// :220
static void ReadsAFileThatCarriesNoMilaConfig()
{
    const std::string path = scratch( "foreign" );
    writeOneFp32( path, { 3.0f, 4.0f } );
    PretrainedModelReader reader( path );
    EXPECT( reader.hasTensor( "w" ) ); EXPECT( reader.getPretrainedMetadata().architecture.empty() );
}

// :242
static void MetadataSurvivesAFullWriteReadCycle()
{
    const std::string path = scratch( "metadata_cycle" );
    PretrainedMetadata o;
    o.architecture = "gemma"; o.model_name = "gemma-4-12b"; o.vocab_size = 262144; o.max_seq_length = 131072; o.embedding_dim = 3840; o.num_layers = 48; o.num_heads = 16;
    o.num_kv_heads = 8; o.head_dim = 256; o.hidden_dim = 15360; o.use_bias = false; o.tie_word_embeddings = true; o.activation = "gelu"; o.norm_type = "rmsnorm";
    o.attention_type = "gqa"; o.positional_encoding = "rope"; o.rope_theta = 1000000.0f; o.norm_epsilon = 1e-6f; o.global_head_dim = 256; o.num_global_kv_heads = 4;
    o.key_equals_value = true; o.window = 1024; o.sliding_window_pattern = 6; o.global_rotary_dim = 128; o.rope_theta_local = 10000.0f; o.rope_theta_global = 1000000.0f;
    o.final_logit_softcapping = 30.0f;
    writeOneFp32( path, { 1.0f }, kMilaConfigMetadataKey, toMetadataJSON( o ) );
    PretrainedModelReader reader( path );
    const auto& r = reader.getPretrainedMetadata();
    EXPECT( r.architecture == o.architecture ); EXPECT( r.model_name == o.model_name ); EXPECT( r.vocab_size == o.vocab_size ); EXPECT( r.max_seq_length == o.max_seq_length );
    EXPECT( r.embedding_dim == o.embedding_dim ); EXPECT( r.num_layers == o.num_layers ); EXPECT( r.num_heads == o.num_heads ); EXPECT( r.num_kv_heads == o.num_kv_heads );
    EXPECT( r.head_dim == o.head_dim ); EXPECT( r.hidden_dim == o.hidden_dim ); EXPECT( r.use_bias == o.use_bias ); EXPECT( r.tie_word_embeddings == o.tie_word_embeddings );
    EXPECT( r.activation == o.activation ); EXPECT( r.norm_type == o.norm_type ); EXPECT( r.attention_type == o.attention_type ); EXPECT( r.positional_encoding == o.positional_encoding );
    EXPECT( floatEq( r.rope_theta, o.rope_theta ) ); EXPECT( floatEq( r.norm_epsilon, o.norm_epsilon ) ); EXPECT( r.global_head_dim == o.global_head_dim );
    EXPECT( r.num_global_kv_heads == o.num_global_kv_heads ); EXPECT( r.key_equals_value == o.key_equals_value ); EXPECT( r.window == o.window );
    EXPECT( r.sliding_window_pattern == o.sliding_window_pattern ); EXPECT( r.global_rotary_dim == o.global_rotary_dim ); EXPECT( floatEq( r.rope_theta_local, o.rope_theta_local ) );
    EXPECT( floatEq( r.rope_theta_global, o.rope_theta_global ) ); EXPECT( floatEq( r.final_logit_softcapping, o.final_logit_softcapping ) );
}

// :320
static void SurfacesTheDeclaredWeightQuantization()
{
    const std::string path = scratch( "quantization" );
    writeOneFp32( path, { 1.0f }, kMilaQuantizationMetadataKey, "per_group_fp4_128" );
    EXPECT( PretrainedModelReader( path ).getWeightQuantization() == "per_group_fp4_128" );
}

// :340
static void TreatsAnUnquantizedDeclarationAsAbsent()
{
    const std::string declared = scratch( "quant_none" ), omitted = scratch( "quant_absent" );
    writeOneFp32( declared, { 1.0f }, kMilaQuantizationMetadataKey, "none" );
    writeOneFp32( omitted, { 1.0f } );
    EXPECT( PretrainedModelReader( declared ).getWeightQuantization().empty() );
    EXPECT( PretrainedModelReader( omitted ).getWeightQuantization().empty() );
}

// :368
static void LegacyMilaContainerDeclaresNoQuantization()
{
    const std::string path = scratch( "legacy_quant" );
    writeMilaFormatFile( path, R"({"architecture":"gpt2"})", "w", { 1.0f, 2.0f } );
    EXPECT( PretrainedModelReader( path ).getWeightQuantization().empty() );
}

// :384
static void RejectsOutOfOrderBodyWrites()
{
    const std::vector<float> b{ 2.0f };
    SafeTensorsWriter writer( scratch( "order" ) );
    writer.declareTensor( "first", "F32", { 1 } );
    writer.declareTensor( "second", "F32", { 1 } );
    writer.beginData();
    EXPECT_THROW_RUNTIME( writer.writeTensorData( "second", b.data(), sizeof( float ) ) );
}

// :402
static void RejectsBodySizeMismatch()
{
    const std::vector<float> values{ 1.0f, 2.0f, 3.0f };
    SafeTensorsWriter writer( scratch( "size" ) );
    writer.declareTensor( "w", "F32", { 2 } );
    writer.beginData();
    EXPECT_THROW_RUNTIME( writer.writeTensorData( "w", values.data(), 3 * sizeof( float ) ) );
}

// :417
static void RejectsDuplicateTensorNames()
{
    SafeTensorsWriter writer( scratch( "duplicate" ) );
    writer.declareTensor( "w", "F32", { 1 } );
    EXPECT_THROW_RUNTIME( writer.declareTensor( "w", "F32", { 1 } ) );
}

// :429
static void RejectsDeclarationAfterHeaderIsWritten()
{
    SafeTensorsWriter writer( scratch( "late" ) );
    writer.declareTensor( "w", "F32", { 1 } );
    writer.beginData();
    EXPECT_THROW_RUNTIME( writer.declareTensor( "late", "F32", { 1 } ) );
}

// :442
static void CloseRefusesWhenADeclaredTensorWasNeverWritten()
{
    const std::vector<float> a{ 1.0f };
    SafeTensorsWriter writer( scratch( "incomplete" ) );
    writer.declareTensor( "first", "F32", { 1 } );
    writer.declareTensor( "second", "F32", { 1 } );
    writer.beginData();
    writer.writeTensorData( "first", a.data(), sizeof( float ) );
    EXPECT_THROW_RUNTIME( writer.close() );
}

// :463
static void RejectsAFileThatIsNeitherContainer()
{
    const std::string path = scratch( "garbage" );
    { std::FILE* f = std::fopen( path.c_str(), "wb" ); const std::string junk( 64, 'x' ); std::fwrite( junk.data(), 1, junk.size(), f ); std::fclose( f ); }
    EXPECT_THROW_RUNTIME( PretrainedModelReader reader( path ) );
}

// :478
static void RejectsATensorExtendingPastEndOfFile()
{
    const std::string path = scratch( "truncated" );
    writeOneFp32( path, { 1.0f, 2.0f, 3.0f, 4.0f } );
    std::FILE* f = std::fopen( path.c_str(), "rb" ); std::fseek( f, 0, SEEK_END ); const long full = std::ftell( f ); std::fclose( f );
    EXPECT( ::truncate( path.c_str(), full - static_cast<long>( sizeof( float ) ) ) == 0 );
    EXPECT_THROW_RUNTIME( PretrainedModelReader reader( path ) );
}

// :502
static void StillReadsTheLegacyMilaContainer()
{
    const std::string path = scratch( "legacy" );
    const std::vector<float> values{ 1.5f, -2.5f, 3.5f, -4.5f, 5.5f, -6.5f };
    const std::string metadata = R"({"architecture":"gpt2","model_name":"legacy","vocab_size":50257,)" R"("embedding_dim":768,"num_layers":12,"num_heads":12})";
    writeMilaFormatFile( path, metadata, "lenc.wte.weight", values );
    PretrainedModelReader reader( path );
    EXPECT( reader.getPretrainedMetadata().architecture == "gpt2" ); EXPECT( reader.getPretrainedMetadata().vocab_size == 50257u ); EXPECT( reader.getPretrainedMetadata().num_layers == 12u );
    EXPECT( reader.hasTensor( "lenc.wte.weight" ) );
    const auto& e = reader.get( "lenc.wte.weight" );
    EXPECT( e.dtype == "F32" ); EXPECT( e.shape.size() == 2 && e.shape[ 0 ] == 2 && e.shape[ 1 ] == 3 );
    EXPECT( e.nbytes() == values.size() * sizeof( float ) ); EXPECT( std::memcmp( e.data, values.data(), e.nbytes() ) == 0 );
}

// :532
static void LegacyContainerStillRejectsAWrongVersion()
{
    const std::string path = scratch( "legacy_version" );
    writeMilaFormatFile( path, R"({"architecture":"gpt2"})", "w", { 1.0f, 2.0f } );
    { std::FILE* f = std::fopen( path.c_str(), "r+b" ); const uint32_t bad_version = 99; std::fseek( f, sizeof( uint32_t ), SEEK_SET ); std::fwrite( &bad_version, 4, 1, f ); std::fclose( f ); }
    EXPECT_THROW_RUNTIME( PretrainedModelReader reader( path ) );
}

int main( int argc, char** argv )
{
    const std::map<std::string, std::function<void()>> scenarios{
        { "RoundTripsTensorsOfMixedDataTypes", RoundTripsTensorsOfMixedDataTypes }, { "CarriesMilaConfigThroughMetadata", CarriesMilaConfigThroughMetadata },
        { "ReadsAFileThatCarriesNoMilaConfig", ReadsAFileThatCarriesNoMilaConfig }, { "MetadataSurvivesAFullWriteReadCycle", MetadataSurvivesAFullWriteReadCycle },
        { "SurfacesTheDeclaredWeightQuantization", SurfacesTheDeclaredWeightQuantization }, { "TreatsAnUnquantizedDeclarationAsAbsent", TreatsAnUnquantizedDeclarationAsAbsent },
        { "LegacyMilaContainerDeclaresNoQuantization", LegacyMilaContainerDeclaresNoQuantization }, { "RejectsOutOfOrderBodyWrites", RejectsOutOfOrderBodyWrites },
        { "RejectsBodySizeMismatch", RejectsBodySizeMismatch }, { "RejectsDuplicateTensorNames", RejectsDuplicateTensorNames },
        { "RejectsDeclarationAfterHeaderIsWritten", RejectsDeclarationAfterHeaderIsWritten }, { "CloseRefusesWhenADeclaredTensorWasNeverWritten", CloseRefusesWhenADeclaredTensorWasNeverWritten },
        { "RejectsAFileThatIsNeitherContainer", RejectsAFileThatIsNeitherContainer }, { "RejectsATensorExtendingPastEndOfFile", RejectsATensorExtendingPastEndOfFile },
        { "StillReadsTheLegacyMilaContainer", StillReadsTheLegacyMilaContainer }, { "LegacyContainerStillRejectsAWrongVersion", LegacyContainerStillRejectsAWrongVersion } };
    if ( argc < 3 ) { for ( auto& [ n, f ] : scenarios ) std::printf( "%s\n", n.c_str() ); return argc == 1 ? 0 : 2; }
    auto it = scenarios.find( argv[ 1 ] );
    if ( it == scenarios.end() ) { std::printf( "unknown scenario %s\n", argv[ 1 ] ); return 2; }
    g_dir = argv[ 2 ];
    try { it->second(); }
    catch ( const std::exception& e ) { std::printf( "  unexpected exception: %s\n", e.what() ); return 1; }
    return g_failures ? 1 : 0;
}
