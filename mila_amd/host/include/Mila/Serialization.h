// Weight ingestion for the hot path: the two containers the reference's PretrainedModelReader accepts
// (Serialization/PretrainedReader.ixx:222-246, sniffed by the leading magic) -- the MILA `.bin` container every converted checkpoint is
// in (MilaBinWriter / PretrainedModelReader below) and the flat SafeTensors container the reference now writes and
// reads its pretrained tensors in -- `<component path>.weight`, `<component path>.weight_scale` siblings for quantized Linears
// (Mila/Src/Dnn/Serialization/SafeTensors.ixx: SafeTensorsWriter declareTensor / setMetadata / beginData / writeTensorData / close;
// Components/Linear/Linear.ixx:370-400, :529-600; PretrainedReader.ixx streams blobs in ascending file order).
//
// The format is the public one: u64 little-endian header length, a JSON object
//   { "<name>": { "dtype": "BF16", "shape": [N, K], "data_offsets": [begin, end] }, ..., "__metadata__": { "k": "v" } }
// padded with spaces to a multiple of 8 bytes, then the raw tensor bytes.  Files written here load in the Python `safetensors`
// package and vice versa (tests/test_serialization_cpu.py).
//
// MI355X-side design: the whole file is memory-mapped and tensors are consumed in ascending offset order (one sequential pass,
// as the reference's streamTensorBlobs); the caller copies a blob host -> device and, for a bf16 blob under a quantized policy,
// quantizes it on the device (Linear::loadParameter) -- 288 GB of HBM and 3 TB/s-class host links make per-tensor staging
// buffers of any size affordable, so there is no chunking logic.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace Mila::Dnn::Serialization
{
    /// dtype names of the container (SafeTensors.ixx: toSafeTensorsDataTypeName / storageBytesPerElement)
    inline size_t safeTensorsElementBytes( const std::string& dtype )
    {
        if ( dtype == "F32" || dtype == "I32" || dtype == "U32" ) return 4;
        if ( dtype == "BF16" || dtype == "F16" || dtype == "I16" || dtype == "U16" ) return 2;
        if ( dtype == "F8_E4M3" || dtype == "F8_E5M2" || dtype == "U8" || dtype == "I8" || dtype == "BOOL" ) return 1;
        if ( dtype == "F64" || dtype == "I64" || dtype == "U64" ) return 8;
        throw std::runtime_error( "SafeTensors: unsupported dtype '" + dtype + "'" );
    }

    struct SafeTensorsEntry
    {
        std::string name, dtype;
        std::vector<int64_t> shape;
        uint64_t begin{ 0 }, end{ 0 };       // byte offsets inside the data section
        const void* data{ nullptr };         // reader only: pointer into the mapped file
        size_t nbytes() const noexcept { return static_cast<size_t>( end - begin ); }
        int64_t elements() const noexcept { int64_t n = 1; for ( auto d : shape ) n *= d; return n; }
    };

    /// Memory-mapped reader.  Throws std::runtime_error on I/O errors and on a malformed container (as the reference's does).
    class SafeTensorsReader
    {
    public:
        explicit SafeTensorsReader( const std::string& path ) : path_( path )
        {
            fd_ = ::open( path.c_str(), O_RDONLY );
            if ( fd_ < 0 ) throw std::runtime_error( "SafeTensorsReader: cannot open '" + path + "'" );
            struct stat st{};
            if ( ::fstat( fd_, &st ) != 0 || st.st_size < 8 ) { ::close( fd_ ); throw std::runtime_error( "SafeTensorsReader: '" + path + "' is too small to be a SafeTensors file" ); }
            size_ = static_cast<size_t>( st.st_size );
            map_ = ::mmap( nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0 );
            if ( map_ == MAP_FAILED ) { ::close( fd_ ); throw std::runtime_error( "SafeTensorsReader: mmap failed for '" + path + "'" ); }
            try { parse(); }
            catch ( ... ) { ::munmap( map_, size_ ); ::close( fd_ ); throw; }
        }
        ~SafeTensorsReader()
        {
            if ( map_ && map_ != MAP_FAILED ) ::munmap( map_, size_ );
            if ( fd_ >= 0 ) ::close( fd_ );
        }
        SafeTensorsReader( const SafeTensorsReader& ) = delete;
        SafeTensorsReader& operator=( const SafeTensorsReader& ) = delete;

        /// entries in ascending file offset order (the order to consume them in)
        const std::vector<SafeTensorsEntry>& entries() const noexcept { return entries_; }
        const std::map<std::string, std::string>& metadata() const noexcept { return metadata_; }
        bool contains( const std::string& name ) const noexcept { return index_.count( name ) != 0; }
        const SafeTensorsEntry& get( const std::string& name ) const
        {
            auto it = index_.find( name );
            if ( it == index_.end() ) throw std::runtime_error( "SafeTensorsReader: '" + path_ + "' has no tensor '" + name + "'" );
            return entries_[ it->second ];
        }

    private:
        // ---- a JSON reader for exactly the header's grammar: objects, arrays, strings, integers ----
        struct Cur { const char* p; const char* e; };
        static void ws( Cur& c ) { while ( c.p < c.e && ( *c.p == ' ' || *c.p == '\n' || *c.p == '\t' || *c.p == '\r' ) ) ++c.p; }
        [[noreturn]] void bad( const char* what ) const { throw std::runtime_error( "SafeTensorsReader: malformed header in '" + path_ + "': " + what ); }
        void expect( Cur& c, char ch ) const { ws( c ); if ( c.p >= c.e || *c.p != ch ) bad( "unexpected character" ); ++c.p; }
        std::string str( Cur& c ) const
        {
            expect( c, '"' );
            std::string s;
            while ( c.p < c.e && *c.p != '"' )
            {
                if ( *c.p == '\\' )
                {
                    if ( ++c.p >= c.e ) bad( "truncated escape" );
                    switch ( *c.p )
                    {
                        case 'n': s += '\n'; break; case 't': s += '\t'; break; case 'r': s += '\r'; break; case 'b': s += '\b'; break; case 'f': s += '\f'; break;
                        case 'u':
                        {
                            if ( c.e - c.p < 5 ) bad( "truncated \\u escape" );
                            unsigned v = 0;
                            for ( int i = 1; i <= 4; ++i ) { const char h = c.p[ i ]; v = v * 16 + ( h >= '0' && h <= '9' ? h - '0' : ( h | 32 ) >= 'a' && ( h | 32 ) <= 'f' ? ( h | 32 ) - 'a' + 10 : 99 ); }
                            if ( v < 0x80 ) s += static_cast<char>( v ); else if ( v < 0x800 ) { s += static_cast<char>( 0xC0 | ( v >> 6 ) ); s += static_cast<char>( 0x80 | ( v & 63 ) ); }
                            else { s += static_cast<char>( 0xE0 | ( v >> 12 ) ); s += static_cast<char>( 0x80 | ( ( v >> 6 ) & 63 ) ); s += static_cast<char>( 0x80 | ( v & 63 ) ); }
                            c.p += 4;
                            break;
                        }
                        default: s += *c.p;
                    }
                    ++c.p;
                }
                else s += *c.p++;
            }
            if ( c.p >= c.e ) bad( "unterminated string" );
            ++c.p;
            return s;
        }
        int64_t integer( Cur& c ) const
        {
            ws( c );
            bool neg = false;
            if ( c.p < c.e && *c.p == '-' ) { neg = true; ++c.p; }
            if ( c.p >= c.e || *c.p < '0' || *c.p > '9' ) bad( "expected an integer" );
            int64_t v = 0;
            while ( c.p < c.e && *c.p >= '0' && *c.p <= '9' ) v = v * 10 + ( *c.p++ - '0' );
            return neg ? -v : v;
        }
        std::vector<int64_t> intArray( Cur& c ) const
        {
            std::vector<int64_t> out;
            expect( c, '[' );
            ws( c );
            if ( c.p < c.e && *c.p == ']' ) { ++c.p; return out; }
            for ( ;; )
            {
                out.push_back( integer( c ) );
                ws( c );
                if ( c.p < c.e && *c.p == ',' ) { ++c.p; continue; }
                expect( c, ']' );
                return out;
            }
        }

        void parse()
        {
            const auto* base = static_cast<const unsigned char*>( map_ );
            uint64_t hlen = 0;
            for ( int i = 7; i >= 0; --i ) hlen = ( hlen << 8 ) | base[ i ];
            if ( hlen == 0 || hlen > size_ - 8 ) bad( "header length exceeds the file" );
            const size_t data0 = 8 + static_cast<size_t>( hlen ), data_bytes = size_ - data0;
            Cur c{ reinterpret_cast<const char*>( base + 8 ), reinterpret_cast<const char*>( base + data0 ) };
            expect( c, '{' );
            ws( c );
            if ( c.p < c.e && *c.p == '}' ) return;
            for ( ;; )
            {
                const std::string key = str( c );
                expect( c, ':' );
                expect( c, '{' );
                if ( key == "__metadata__" )
                {
                    ws( c );
                    if ( c.p < c.e && *c.p == '}' ) ++c.p;
                    else
                        for ( ;; )
                        {
                            const std::string k = str( c );
                            expect( c, ':' );
                            metadata_[ k ] = str( c );
                            ws( c );
                            if ( c.p < c.e && *c.p == ',' ) { ++c.p; continue; }
                            expect( c, '}' );
                            break;
                        }
                }
                else
                {
                    SafeTensorsEntry e;
                    e.name = key;
                    bool have_off = false;
                    for ( ;; )
                    {
                        const std::string f = str( c );
                        expect( c, ':' );
                        if ( f == "dtype" ) e.dtype = str( c );
                        else if ( f == "shape" ) e.shape = intArray( c );
                        else if ( f == "data_offsets" )
                        {
                            const auto o = intArray( c );
                            if ( o.size() != 2 || o[ 0 ] < 0 || o[ 1 ] < o[ 0 ] ) bad( "data_offsets must be [begin, end]" );
                            e.begin = static_cast<uint64_t>( o[ 0 ] ); e.end = static_cast<uint64_t>( o[ 1 ] );
                            have_off = true;
                        }
                        else bad( "unknown tensor field" );
                        ws( c );
                        if ( c.p < c.e && *c.p == ',' ) { ++c.p; continue; }
                        expect( c, '}' );
                        break;
                    }
                    if ( e.dtype.empty() || !have_off ) bad( "tensor entry without dtype or data_offsets" );
                    if ( e.end > data_bytes ) bad( "tensor data runs past the end of the file" );
                    for ( auto d : e.shape ) if ( d < 0 ) bad( "negative dimension" );
                    // sub-byte storage (packed fp4) is carried as U8 by the writer, so every dtype here has whole bytes per element
                    if ( static_cast<uint64_t>( e.elements() ) * safeTensorsElementBytes( e.dtype ) != e.end - e.begin ) bad( "shape x dtype does not match data_offsets" );
                    e.data = base + data0 + e.begin;
                    entries_.push_back( std::move( e ) );
                }
                ws( c );
                if ( c.p < c.e && *c.p == ',' ) { ++c.p; continue; }
                expect( c, '}' );
                break;
            }
            std::sort( entries_.begin(), entries_.end(), []( const SafeTensorsEntry& a, const SafeTensorsEntry& b ) { return a.begin < b.begin; } );
            for ( size_t i = 0; i < entries_.size(); ++i )
            {
                if ( i && entries_[ i ].begin < entries_[ i - 1 ].end ) bad( "overlapping tensors" );
                if ( !index_.emplace( entries_[ i ].name, i ).second ) bad( "duplicate tensor name" );
            }
        }

        std::string path_;
        int fd_{ -1 };
        void* map_{ nullptr };
        size_t size_{ 0 };
        std::vector<SafeTensorsEntry> entries_;
        std::map<std::string, size_t> index_;
        std::map<std::string, std::string> metadata_;
    };

    /// Two-pass writer with the reference writer's call sequence (SafeTensors.ixx:165-400): declareTensor()* -> setMetadata()* ->
    /// beginData() -> writeTensorData() in declaration order -> close().
    class SafeTensorsWriter
    {
    public:
        explicit SafeTensorsWriter( const std::string& path ) : path_( path )
        {
            file_ = std::fopen( path.c_str(), "wb" );
            if ( !file_ ) throw std::runtime_error( "SafeTensorsWriter: cannot create '" + path + "'" );
        }
        ~SafeTensorsWriter() { if ( file_ ) std::fclose( file_ ); }
        SafeTensorsWriter( const SafeTensorsWriter& ) = delete;
        SafeTensorsWriter& operator=( const SafeTensorsWriter& ) = delete;

        void declareTensor( const std::string& name, const std::string& dtype, const std::vector<int64_t>& shape )
        {
            if ( header_written_ ) throw std::runtime_error( "SafeTensorsWriter: declareTensor after beginData" );
            for ( auto& e : entries_ ) if ( e.name == name ) throw std::runtime_error( "SafeTensorsWriter: duplicate tensor '" + name + "'" );
            SafeTensorsEntry e;
            e.name = name; e.dtype = dtype; e.shape = shape;
            for ( auto d : shape ) if ( d < 0 ) throw std::runtime_error( "SafeTensorsWriter: negative dimension in '" + name + "'" );
            e.begin = next_offset_;
            e.end = e.begin + static_cast<uint64_t>( e.elements() ) * safeTensorsElementBytes( dtype );
            next_offset_ = e.end;
            entries_.push_back( std::move( e ) );
        }
        size_t getTensorCount() const noexcept { return entries_.size(); }
        void setMetadata( const std::string& key, const std::string& value )
        {
            if ( header_written_ ) throw std::runtime_error( "SafeTensorsWriter: setMetadata after beginData" );
            metadata_[ key ] = value;
        }
        void beginData()
        {
            if ( header_written_ ) throw std::runtime_error( "SafeTensorsWriter: beginData called twice" );
            std::string h = "{";
            if ( !metadata_.empty() )
            {
                h += "\"__metadata__\":{";
                bool first = true;
                for ( auto& [ k, v ] : metadata_ ) { if ( !first ) h += ","; first = false; h += quote( k ) + ":" + quote( v ); }
                h += "},";
            }
            bool first = true;
            for ( auto& e : entries_ )
            {
                if ( !first ) h += ",";
                first = false;
                h += quote( e.name ) + ":{\"dtype\":" + quote( e.dtype ) + ",\"shape\":[";
                for ( size_t d = 0; d < e.shape.size(); ++d ) { if ( d ) h += ","; h += std::to_string( e.shape[ d ] ); }
                h += "],\"data_offsets\":[" + std::to_string( e.begin ) + "," + std::to_string( e.end ) + "]}";
            }
            h += "}";
            while ( ( 8 + h.size() ) % 8 != 0 ) h += ' ';
            unsigned char len[ 8 ];
            uint64_t n = h.size();
            for ( int i = 0; i < 8; ++i ) { len[ i ] = static_cast<unsigned char>( n & 0xff ); n >>= 8; }
            writeExact( len, 8 );
            writeExact( h.data(), h.size() );
            header_written_ = true;
        }
        void writeTensorData( const std::string& name, const void* data, size_t nbytes )
        {
            if ( !header_written_ ) throw std::runtime_error( "SafeTensorsWriter: writeTensorData before beginData" );
            if ( next_write_ >= entries_.size() ) throw std::runtime_error( "SafeTensorsWriter: more tensors written than declared" );
            const auto& e = entries_[ next_write_ ];
            if ( e.name != name ) throw std::runtime_error( "SafeTensorsWriter: tensors must be written in declaration order (expected '" + e.name + "', got '" + name + "')" );
            if ( nbytes != e.nbytes() ) throw std::runtime_error( "SafeTensorsWriter: '" + name + "' has " + std::to_string( nbytes ) + " bytes, declared " + std::to_string( e.nbytes() ) );
            writeExact( data, nbytes );
            ++next_write_;
        }
        void close()
        {
            if ( !file_ ) return;
            if ( header_written_ && next_write_ != entries_.size() ) { std::fclose( file_ ); file_ = nullptr; throw std::runtime_error( "SafeTensorsWriter: close() before every declared tensor was written" ); }
            if ( std::fclose( file_ ) != 0 ) { file_ = nullptr; throw std::runtime_error( "SafeTensorsWriter: closing '" + path_ + "' failed" ); }
            file_ = nullptr;
        }

    private:
        static std::string quote( const std::string& s )
        {
            std::string q = "\"";
            for ( char ch : s )
            {
                if ( ch == '"' || ch == '\\' ) { q += '\\'; q += ch; }
                else if ( static_cast<unsigned char>( ch ) < 0x20 ) { char b[ 8 ]; std::snprintf( b, sizeof b, "\\u%04x", ch ); q += b; }
                else q += ch;
            }
            return q + "\"";
        }
        void writeExact( const void* p, size_t n )
        {
            if ( n && std::fwrite( p, 1, n, file_ ) != n ) throw std::runtime_error( "SafeTensorsWriter: short write to '" + path_ + "'" );
        }
        std::string path_;
        std::FILE* file_{ nullptr };
        std::vector<SafeTensorsEntry> entries_;
        std::map<std::string, std::string> metadata_;
        uint64_t next_offset_{ 0 };
        size_t next_write_{ 0 };
        bool header_written_{ false };
    };

    // -------------------------------------------------------------------------------------------------------------------------
    // Model description carried by both containers (PretrainedReader.ixx:96-128 PretrainedMetadata, :131-175 toMetadataJSON,
    // :1160-1251 parseMetadataJSON): in a MILA .bin it is the JSON block after the header, in a SafeTensors file the string under
    // __metadata__["mila_config"]; __metadata__["mila_quantization"] names the policy of a pre-quantized artifact.
    // -------------------------------------------------------------------------------------------------------------------------
    inline constexpr const char* kMilaConfigMetadataKey = "mila_config";
    inline constexpr const char* kMilaQuantizationMetadataKey = "mila_quantization";

    struct PretrainedMetadata
    {
        std::string architecture, model_name;
        uint32_t vocab_size{ 0 }, max_seq_length{ 0 }, embedding_dim{ 0 }, num_layers{ 0 }, num_heads{ 0 }, num_kv_heads{ 0 }, head_dim{ 0 }, hidden_dim{ 0 };
        bool use_bias{ false }, tie_word_embeddings{ false };
        std::string activation, norm_type, attention_type, positional_encoding;
        float rope_theta{ 0.0f }, norm_epsilon{ 0.0f };
        uint32_t global_head_dim{ 0 }, num_global_kv_heads{ 0 };
        bool key_equals_value{ false };
        uint32_t window{ 0 }, sliding_window_pattern{ 0 }, global_rotary_dim{ 0 };
        float rope_theta_local{ 0.0f }, rope_theta_global{ 0.0f }, final_logit_softcapping{ 0.0f };
    };

    namespace detail
    {
        inline std::string jsonQuote( const std::string& s )
        {
            std::string q = "\"";
            for ( char ch : s )
            {
                if ( ch == '"' || ch == '\\' ) { q += '\\'; q += ch; }
                else if ( static_cast<unsigned char>( ch ) < 0x20 ) { char b[ 8 ]; std::snprintf( b, sizeof b, "\\u%04x", ch ); q += b; }
                else q += ch;
            }
            return q + "\"";
        }
        inline std::string jsonFloat( float v ) { char b[ 48 ]; std::snprintf( b, sizeof b, "%.9g", static_cast<double>( v ) ); return b; }
        /// the value text that follows `"key":` up to the next ',' or '}' outside a string; empty when the key is absent.  Keys are
        /// matched with both quotes, so "rope_theta" does not match inside "rope_theta_local" (the reference parser's rule)
        inline std::string jsonValueText( const std::string& json, const std::string& key )
        {
            const std::string k = "\"" + key + "\"";
            size_t pos = 0;
            for ( ;; )
            {
                pos = json.find( k, pos );
                if ( pos == std::string::npos ) return "";
                size_t c = pos + k.size();
                while ( c < json.size() && ( json[ c ] == ' ' || json[ c ] == '\t' || json[ c ] == '\n' || json[ c ] == '\r' ) ) ++c;
                if ( c < json.size() && json[ c ] == ':' ) { pos = c + 1; break; }
                pos += k.size();     // the text was a string VALUE, not a key
            }
            while ( pos < json.size() && ( json[ pos ] == ' ' || json[ pos ] == '\t' || json[ pos ] == '\n' || json[ pos ] == '\r' ) ) ++pos;
            size_t e = pos;
            if ( e < json.size() && json[ e ] == '"' )
            {
                for ( ++e; e < json.size() && json[ e ] != '"'; ++e ) if ( json[ e ] == '\\' ) ++e;
                return json.substr( pos, std::min( e + 1, json.size() ) - pos );
            }
            while ( e < json.size() && json[ e ] != ',' && json[ e ] != '}' ) ++e;
            while ( e > pos && ( json[ e - 1 ] == ' ' || json[ e - 1 ] == '\n' || json[ e - 1 ] == '\t' || json[ e - 1 ] == '\r' ) ) --e;
            return json.substr( pos, e - pos );
        }
    }

    inline std::string toMetadataJSON( const PretrainedMetadata& m )
    {
        using detail::jsonQuote; using detail::jsonFloat;
        auto b = []( bool v ) { return std::string( v ? "true" : "false" ); };
        std::string j = "{";
        j += "\"architecture\":" + jsonQuote( m.architecture ) + ",\"model_name\":" + jsonQuote( m.model_name );
        j += ",\"vocab_size\":" + std::to_string( m.vocab_size ) + ",\"max_seq_length\":" + std::to_string( m.max_seq_length );
        j += ",\"embedding_dim\":" + std::to_string( m.embedding_dim ) + ",\"num_layers\":" + std::to_string( m.num_layers );
        j += ",\"num_heads\":" + std::to_string( m.num_heads ) + ",\"num_kv_heads\":" + std::to_string( m.num_kv_heads );
        j += ",\"head_dim\":" + std::to_string( m.head_dim ) + ",\"hidden_dim\":" + std::to_string( m.hidden_dim );
        j += ",\"use_bias\":" + b( m.use_bias ) + ",\"tie_word_embeddings\":" + b( m.tie_word_embeddings );
        j += ",\"activation\":" + jsonQuote( m.activation ) + ",\"norm_type\":" + jsonQuote( m.norm_type );
        j += ",\"attention_type\":" + jsonQuote( m.attention_type ) + ",\"positional_encoding\":" + jsonQuote( m.positional_encoding );
        j += ",\"rope_theta\":" + jsonFloat( m.rope_theta ) + ",\"norm_epsilon\":" + jsonFloat( m.norm_epsilon );
        j += ",\"global_head_dim\":" + std::to_string( m.global_head_dim ) + ",\"num_global_kv_heads\":" + std::to_string( m.num_global_kv_heads );
        j += ",\"key_equals_value\":" + b( m.key_equals_value ) + ",\"window\":" + std::to_string( m.window );
        j += ",\"sliding_window_pattern\":" + std::to_string( m.sliding_window_pattern ) + ",\"global_rotary_dim\":" + std::to_string( m.global_rotary_dim );
        j += ",\"rope_theta_local\":" + jsonFloat( m.rope_theta_local ) + ",\"rope_theta_global\":" + jsonFloat( m.rope_theta_global );
        j += ",\"final_logit_softcapping\":" + jsonFloat( m.final_logit_softcapping ) + "}";
        return j;
    }

    /// absent keys leave the zero / false / empty defaults, like the reference parser
    inline PretrainedMetadata parseMetadataJSON( const std::string& json )
    {
        auto str = [&]( const char* k ) { std::string v = detail::jsonValueText( json, k ); return ( v.size() >= 2 && v.front() == '"' && v.back() == '"' ) ? v.substr( 1, v.size() - 2 ) : std::string(); };
        auto u32 = [&]( const char* k ) -> uint32_t { const std::string v = detail::jsonValueText( json, k ); try { return v.empty() ? 0u : static_cast<uint32_t>( std::stoul( v ) ); } catch ( ... ) { return 0u; } };
        auto f32 = [&]( const char* k ) -> float { const std::string v = detail::jsonValueText( json, k ); try { return v.empty() ? 0.0f : std::stof( v ); } catch ( ... ) { return 0.0f; } };
        auto flag = [&]( const char* k ) { return detail::jsonValueText( json, k ) == "true"; };
        PretrainedMetadata m;
        m.architecture = str( "architecture" ); m.model_name = str( "model_name" );
        m.vocab_size = u32( "vocab_size" ); m.max_seq_length = u32( "max_seq_length" ); m.embedding_dim = u32( "embedding_dim" ); m.num_layers = u32( "num_layers" );
        m.num_heads = u32( "num_heads" ); m.num_kv_heads = u32( "num_kv_heads" ); m.head_dim = u32( "head_dim" ); m.hidden_dim = u32( "hidden_dim" );
        m.use_bias = flag( "use_bias" ); m.tie_word_embeddings = flag( "tie_word_embeddings" );
        m.activation = str( "activation" ); m.norm_type = str( "norm_type" ); m.attention_type = str( "attention_type" ); m.positional_encoding = str( "positional_encoding" );
        m.rope_theta = f32( "rope_theta" ); m.norm_epsilon = f32( "norm_epsilon" );
        m.global_head_dim = u32( "global_head_dim" ); m.num_global_kv_heads = u32( "num_global_kv_heads" ); m.key_equals_value = flag( "key_equals_value" );
        m.window = u32( "window" ); m.sliding_window_pattern = u32( "sliding_window_pattern" ); m.global_rotary_dim = u32( "global_rotary_dim" );
        m.rope_theta_local = f32( "rope_theta_local" ); m.rope_theta_global = f32( "rope_theta_global" ); m.final_logit_softcapping = f32( "final_logit_softcapping" );
        return m;
    }

    // -------------------------------------------------------------------------------------------------------------------------
    // The MILA .bin container (PretrainedReader.ixx:222-231, :1121-1306): every field little-endian,
    //   u32 magic 0x4D494C41 | u32 version 1 | u32 num_tensors | u32 metadata bytes | metadata JSON |
    //   num_tensors x { u32 name bytes | name | u32 dtype code | u32 rank | rank x u32 extent | u64 ABSOLUTE file offset | u64 nbytes } | tensor bytes
    // dtype wire codes (:182-191): 0 F32, 1 F16, 2 BF16, 3 I32 are the original set; 4 U8, 5 F8_E4M3, 6 F8_E5M2, 7 I8.
    // -------------------------------------------------------------------------------------------------------------------------
    inline constexpr uint32_t kMilaBinMagic = 0x4D494C41u;
    inline constexpr uint32_t kMilaBinVersion = 1u;
    inline constexpr uint32_t kMilaBinMaxRank = 8u;

    inline const char* milaWireCodeToDtypeName( uint32_t code )
    {
        static const char* names[] = { "F32", "F16", "BF16", "I32", "U8", "F8_E4M3", "F8_E5M2", "I8" };
        if ( code >= 8 ) throw std::runtime_error( "MILA container: unknown dtype code " + std::to_string( code ) );
        return names[ code ];
    }
    inline uint32_t dtypeNameToMilaWireCode( const std::string& dtype )
    {
        static const char* names[] = { "F32", "F16", "BF16", "I32", "U8", "F8_E4M3", "F8_E5M2", "I8" };
        for ( uint32_t i = 0; i < 8; ++i ) if ( dtype == names[ i ] ) return i;
        throw std::runtime_error( "MILA container: dtype '" + dtype + "' has no wire code" );
    }

    /// Writer with the SafeTensorsWriter's call sequence (declareTensor* -> setMetadataJSON -> beginData -> writeTensorData in declaration
    /// order -> close), so one save routine drives either container.  The reference itself only READS this container (its writer is the
    /// Python converter); this one exists so that the reader is tested against files laid out byte for byte as that description says.
    class MilaBinWriter
    {
    public:
        explicit MilaBinWriter( const std::string& path ) : path_( path )
        {
            file_ = std::fopen( path.c_str(), "wb" );
            if ( !file_ ) throw std::runtime_error( "MilaBinWriter: cannot create '" + path + "'" );
        }
        ~MilaBinWriter() { if ( file_ ) std::fclose( file_ ); }
        MilaBinWriter( const MilaBinWriter& ) = delete;
        MilaBinWriter& operator=( const MilaBinWriter& ) = delete;

        void declareTensor( const std::string& name, const std::string& dtype, const std::vector<int64_t>& shape )
        {
            if ( header_written_ ) throw std::runtime_error( "MilaBinWriter: declareTensor after beginData" );
            if ( name.empty() || name.size() > 1024 ) throw std::runtime_error( "MilaBinWriter: tensor names are 1..1024 bytes" );
            if ( shape.size() > kMilaBinMaxRank ) throw std::runtime_error( "MilaBinWriter: rank above " + std::to_string( kMilaBinMaxRank ) );
            for ( auto& e : entries_ ) if ( e.name == name ) throw std::runtime_error( "MilaBinWriter: duplicate tensor '" + name + "'" );
            SafeTensorsEntry e;
            e.name = name; e.dtype = dtype; e.shape = shape;
            (void)dtypeNameToMilaWireCode( dtype );
            for ( auto d : shape ) if ( d < 0 || d > 0xffffffffLL ) throw std::runtime_error( "MilaBinWriter: extent of '" + name + "' does not fit 32 bits" );
            e.begin = next_offset_;
            e.end = e.begin + static_cast<uint64_t>( e.elements() ) * safeTensorsElementBytes( dtype );
            next_offset_ = e.end;
            entries_.push_back( std::move( e ) );
        }
        void setMetadataJSON( const std::string& json )
        {
            if ( header_written_ ) throw std::runtime_error( "MilaBinWriter: setMetadataJSON after beginData" );
            metadata_json_ = json;
        }
        void beginData()
        {
            if ( header_written_ ) throw std::runtime_error( "MilaBinWriter: beginData called twice" );
            if ( metadata_json_.empty() ) throw std::runtime_error( "MilaBinWriter: the container requires a metadata block" );
            uint64_t data0 = 16 + metadata_json_.size();
            for ( auto& e : entries_ ) data0 += 4 + e.name.size() + 4 + 4 + 4 * e.shape.size() + 8 + 8;
            u32( kMilaBinMagic ); u32( kMilaBinVersion ); u32( static_cast<uint32_t>( entries_.size() ) );
            u32( static_cast<uint32_t>( metadata_json_.size() ) );
            writeExact( metadata_json_.data(), metadata_json_.size() );
            for ( auto& e : entries_ )
            {
                u32( static_cast<uint32_t>( e.name.size() ) ); writeExact( e.name.data(), e.name.size() );
                u32( dtypeNameToMilaWireCode( e.dtype ) ); u32( static_cast<uint32_t>( e.shape.size() ) );
                for ( auto d : e.shape ) u32( static_cast<uint32_t>( d ) );
                u64( data0 + e.begin ); u64( e.end - e.begin );
            }
            header_written_ = true;
        }
        void writeTensorData( const std::string& name, const void* data, size_t nbytes )
        {
            if ( !header_written_ ) throw std::runtime_error( "MilaBinWriter: writeTensorData before beginData" );
            if ( next_write_ >= entries_.size() ) throw std::runtime_error( "MilaBinWriter: more tensors written than declared" );
            const auto& e = entries_[ next_write_ ];
            if ( e.name != name ) throw std::runtime_error( "MilaBinWriter: tensors must be written in declaration order (expected '" + e.name + "', got '" + name + "')" );
            if ( nbytes != e.nbytes() ) throw std::runtime_error( "MilaBinWriter: '" + name + "' has " + std::to_string( nbytes ) + " bytes, declared " + std::to_string( e.nbytes() ) );
            writeExact( data, nbytes );
            ++next_write_;
        }
        void close()
        {
            if ( !file_ ) return;
            if ( header_written_ && next_write_ != entries_.size() ) { std::fclose( file_ ); file_ = nullptr; throw std::runtime_error( "MilaBinWriter: close() before every declared tensor was written" ); }
            if ( std::fclose( file_ ) != 0 ) { file_ = nullptr; throw std::runtime_error( "MilaBinWriter: closing '" + path_ + "' failed" ); }
            file_ = nullptr;
        }
    private:
        void u32( uint32_t v ) { unsigned char b[ 4 ]; for ( int i = 0; i < 4; ++i ) { b[ i ] = static_cast<unsigned char>( v & 0xff ); v >>= 8; } writeExact( b, 4 ); }
        void u64( uint64_t v ) { unsigned char b[ 8 ]; for ( int i = 0; i < 8; ++i ) { b[ i ] = static_cast<unsigned char>( v & 0xff ); v >>= 8; } writeExact( b, 8 ); }
        void writeExact( const void* p, size_t n ) { if ( n && std::fwrite( p, 1, n, file_ ) != n ) throw std::runtime_error( "MilaBinWriter: short write to '" + path_ + "'" ); }
        std::string path_, metadata_json_;
        std::FILE* file_{ nullptr };
        std::vector<SafeTensorsEntry> entries_;
        uint64_t next_offset_{ 0 };
        size_t next_write_{ 0 };
        bool header_written_{ false };
    };

    /// Counterpart of PretrainedModelReader (PretrainedReader.ixx:262-296): opens either container, sniffed by the leading magic, and
    /// fills ONE tensor index; entries() is in ascending file-offset order, the order streamTensorBlobs consumes them in (:453-470), so
    /// loading is a single sequential pass over the mapping.  Every failure -- a malformed file, a writer call out of sequence, an I/O error -- throws std::runtime_error, as every throw of the reference's readers and writers does (SafeTensors.ixx, PretrainedReader.ixx).
    class PretrainedModelReader
    {
    public:
        explicit PretrainedModelReader( const std::string& path ) : path_( path )
        {
            {
                std::FILE* f = std::fopen( path.c_str(), "rb" );
                if ( !f ) throw std::runtime_error( "PretrainedModelReader: cannot open pretrained model file '" + path + "'" );
                unsigned char m[ 4 ] = { 0, 0, 0, 0 };
                const size_t got = std::fread( m, 1, 4, f );
                std::fclose( f );
                is_mila_ = got == 4 && ( static_cast<uint32_t>( m[ 0 ] ) | ( static_cast<uint32_t>( m[ 1 ] ) << 8 ) | ( static_cast<uint32_t>( m[ 2 ] ) << 16 ) | ( static_cast<uint32_t>( m[ 3 ] ) << 24 ) ) == kMilaBinMagic;
            }
            if ( !is_mila_ )
            {
                st_ = std::make_unique<SafeTensorsReader>( path );
                entries_ = st_->entries();
                auto q = st_->metadata().find( kMilaQuantizationMetadataKey );
                if ( q != st_->metadata().end() && q->second != "none" ) weight_quantization_ = q->second;
                auto c = st_->metadata().find( kMilaConfigMetadataKey );
                if ( c != st_->metadata().end() ) { metadata_json_ = c->second; metadata_ = parseMetadataJSON( c->second ); }
            }
            else openMila();
            for ( size_t i = 0; i < entries_.size(); ++i ) index_[ entries_[ i ].name ] = i;
        }
        ~PretrainedModelReader()
        {
            if ( map_ && map_ != MAP_FAILED ) ::munmap( map_, size_ );
            if ( fd_ >= 0 ) ::close( fd_ );
        }
        PretrainedModelReader( const PretrainedModelReader& ) = delete;
        PretrainedModelReader& operator=( const PretrainedModelReader& ) = delete;

        bool isMilaContainer() const noexcept { return is_mila_; }
        const PretrainedMetadata& getPretrainedMetadata() const noexcept { return metadata_; }
        const std::string& metadataJSON() const noexcept { return metadata_json_; }
        /// empty = "quantize on load" (a MILA .bin and a BF16 safetensors file both say that, PretrainedReader.ixx:330-344)
        const std::string& getWeightQuantization() const noexcept { return weight_quantization_; }
        const std::vector<SafeTensorsEntry>& entries() const noexcept { return entries_; }
        bool hasTensor( const std::string& name ) const noexcept { return index_.count( name ) != 0; }
        std::vector<std::string> getTensorNames() const { std::vector<std::string> n; for ( auto& e : entries_ ) n.push_back( e.name ); return n; }
        const SafeTensorsEntry& get( const std::string& name ) const
        {
            auto it = index_.find( name );
            if ( it == index_.end() ) throw std::runtime_error( "PretrainedModelReader: '" + path_ + "' has no tensor '" + name + "'" );
            return entries_[ it->second ];
        }
        size_t getMaxTensorSizeBytes() const noexcept { size_t m = 0; for ( auto& e : entries_ ) m = std::max( m, e.nbytes() ); return m; }
        /// every blob once, ascending offsets: consume( name, entry )
        template<typename F> void streamTensorBlobs( F&& consume ) const { for ( auto& e : entries_ ) consume( e.name, e ); }

    private:
        [[noreturn]] void bad( const std::string& what ) const { throw std::runtime_error( "PretrainedModelReader: malformed MILA container '" + path_ + "': " + what ); }
        void openMila()
        {
            fd_ = ::open( path_.c_str(), O_RDONLY );
            if ( fd_ < 0 ) throw std::runtime_error( "PretrainedModelReader: cannot open '" + path_ + "'" );
            struct stat st{};
            if ( ::fstat( fd_, &st ) != 0 || st.st_size < 16 ) bad( "shorter than its fixed header" );
            size_ = static_cast<size_t>( st.st_size );
            map_ = ::mmap( nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0 );
            if ( map_ == MAP_FAILED ) { map_ = nullptr; throw std::runtime_error( "PretrainedModelReader: mmap failed for '" + path_ + "'" ); }
            const auto* base = static_cast<const unsigned char*>( map_ );
            size_t pos = 0;
            auto need = [&]( size_t n, const char* what ) { if ( n > size_ - pos ) bad( std::string( "truncated while reading " ) + what ); };
            auto u32 = [&]( const char* what ) { need( 4, what ); uint32_t v = 0; for ( int i = 3; i >= 0; --i ) v = ( v << 8 ) | base[ pos + i ]; pos += 4; return v; };
            auto u64 = [&]( const char* what ) { need( 8, what ); uint64_t v = 0; for ( int i = 7; i >= 0; --i ) v = ( v << 8 ) | base[ pos + i ]; pos += 8; return v; };
            if ( u32( "magic" ) != kMilaBinMagic ) bad( "wrong magic number" );
            const uint32_t version = u32( "version" );
            if ( version != kMilaBinVersion ) bad( "unsupported file version " + std::to_string( version ) );
            const uint32_t count = u32( "tensor count" );
            const uint32_t mbytes = u32( "metadata size" );
            if ( mbytes == 0 ) bad( "empty metadata block" );
            need( mbytes, "metadata JSON" );
            metadata_json_.assign( reinterpret_cast<const char*>( base + pos ), mbytes );
            pos += mbytes;
            metadata_ = parseMetadataJSON( metadata_json_ );
            // every index record is at least 28 bytes: a count the file cannot hold is rejected before any allocation
            if ( static_cast<uint64_t>( count ) * 28u > size_ - pos ) bad( "tensor count exceeds what the file can hold" );
            entries_.reserve( count );
            for ( uint32_t i = 0; i < count; ++i )
            {
                SafeTensorsEntry e;
                const uint32_t nlen = u32( "tensor name length" );
                if ( nlen == 0 || nlen > 1024 ) bad( "invalid tensor name length at index " + std::to_string( i ) );
                need( nlen, "tensor name" );
                e.name.assign( reinterpret_cast<const char*>( base + pos ), nlen );
                pos += nlen;
                e.dtype = milaWireCodeToDtypeName( u32( "dtype" ) );
                const uint32_t rank = u32( "rank" );
                if ( rank > kMilaBinMaxRank ) bad( "invalid tensor dimensionality for '" + e.name + "'" );
                for ( uint32_t d = 0; d < rank; ++d ) e.shape.push_back( static_cast<int64_t>( u32( "extent" ) ) );
                e.begin = u64( "offset" );
                const uint64_t nbytes = u64( "byte count" );
                if ( e.begin > size_ || nbytes > size_ - e.begin ) bad( "tensor '" + e.name + "' extends past end of file" );
                e.end = e.begin + nbytes;
                // the shape must describe exactly the bytes on disk (extents up to 2^32 each: multiply with an overflow guard)
                uint64_t elems = 1;
                for ( auto d : e.shape ) { if ( d != 0 && elems > ( ~0ull ) / static_cast<uint64_t>( d ) ) bad( "shape of '" + e.name + "' overflows" ); elems *= static_cast<uint64_t>( d ); }
                const uint64_t eb = safeTensorsElementBytes( e.dtype );
                if ( elems > ( ~0ull ) / eb || elems * eb != nbytes ) bad( "shape x dtype of '" + e.name + "' does not match its byte count" );
                e.data = base + e.begin;
                entries_.push_back( std::move( e ) );
            }
            const size_t data0 = pos;
            std::sort( entries_.begin(), entries_.end(), []( const SafeTensorsEntry& a, const SafeTensorsEntry& b ) { return a.begin < b.begin; } );
            std::map<std::string, int> seen;
            for ( size_t i = 0; i < entries_.size(); ++i )
            {
                if ( entries_[ i ].begin < data0 ) bad( "tensor '" + entries_[ i ].name + "' overlaps the header" );
                if ( i && entries_[ i ].begin < entries_[ i - 1 ].end ) bad( "overlapping tensors" );
                if ( seen[ entries_[ i ].name ]++ ) bad( "duplicate tensor name '" + entries_[ i ].name + "'" );
            }
        }

        std::string path_, metadata_json_, weight_quantization_;
        bool is_mila_{ false };
        std::unique_ptr<SafeTensorsReader> st_;
        int fd_{ -1 };
        void* map_{ nullptr };
        size_t size_{ 0 };
        PretrainedMetadata metadata_;
        std::vector<SafeTensorsEntry> entries_;
        std::map<std::string, size_t> index_;
    };
}
