// Weight ingestion for the hot path: the flat SafeTensors container the reference writes and
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
        throw std::invalid_argument( "SafeTensors: unsupported dtype '" + dtype + "'" );
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

    /// Memory-mapped reader.  Throws std::runtime_error on I/O errors and std::invalid_argument on a malformed container.
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
            if ( it == index_.end() ) throw std::invalid_argument( "SafeTensorsReader: '" + path_ + "' has no tensor '" + name + "'" );
            return entries_[ it->second ];
        }

    private:
        // ---- a JSON reader for exactly the header's grammar: objects, arrays, strings, integers ----
        struct Cur { const char* p; const char* e; };
        static void ws( Cur& c ) { while ( c.p < c.e && ( *c.p == ' ' || *c.p == '\n' || *c.p == '\t' || *c.p == '\r' ) ) ++c.p; }
        [[noreturn]] void bad( const char* what ) const { throw std::invalid_argument( "SafeTensorsReader: malformed header in '" + path_ + "': " + what ); }
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
            if ( header_written_ ) throw std::logic_error( "SafeTensorsWriter: declareTensor after beginData" );
            for ( auto& e : entries_ ) if ( e.name == name ) throw std::invalid_argument( "SafeTensorsWriter: duplicate tensor '" + name + "'" );
            SafeTensorsEntry e;
            e.name = name; e.dtype = dtype; e.shape = shape;
            for ( auto d : shape ) if ( d < 0 ) throw std::invalid_argument( "SafeTensorsWriter: negative dimension in '" + name + "'" );
            e.begin = next_offset_;
            e.end = e.begin + static_cast<uint64_t>( e.elements() ) * safeTensorsElementBytes( dtype );
            next_offset_ = e.end;
            entries_.push_back( std::move( e ) );
        }
        void setMetadata( const std::string& key, const std::string& value )
        {
            if ( header_written_ ) throw std::logic_error( "SafeTensorsWriter: setMetadata after beginData" );
            metadata_[ key ] = value;
        }
        void beginData()
        {
            if ( header_written_ ) throw std::logic_error( "SafeTensorsWriter: beginData called twice" );
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
            if ( !header_written_ ) throw std::logic_error( "SafeTensorsWriter: writeTensorData before beginData" );
            if ( next_write_ >= entries_.size() ) throw std::logic_error( "SafeTensorsWriter: more tensors written than declared" );
            const auto& e = entries_[ next_write_ ];
            if ( e.name != name ) throw std::logic_error( "SafeTensorsWriter: tensors must be written in declaration order (expected '" + e.name + "', got '" + name + "')" );
            if ( nbytes != e.nbytes() ) throw std::invalid_argument( "SafeTensorsWriter: '" + name + "' has " + std::to_string( nbytes ) + " bytes, declared " + std::to_string( e.nbytes() ) );
            writeExact( data, nbytes );
            ++next_write_;
        }
        void close()
        {
            if ( !file_ ) return;
            if ( header_written_ && next_write_ != entries_.size() ) { std::fclose( file_ ); file_ = nullptr; throw std::logic_error( "SafeTensorsWriter: close() before every declared tensor was written" ); }
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
}
