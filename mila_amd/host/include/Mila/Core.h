// Mila host-side template surface, restated for a toolchain without C++23 modules / <format>
// (SURVEY.md section 7): DeviceType, TensorDataType, Tensor<dtype, MR>, ExecutionContext<Device>,
// BuildContext.  Same names, template axes and error behaviour as the reference so that component
// code and tests read like the reference's own; DeviceType::Rocm is the working CDNA4 device.
//
// Reference interfaces mirrored (paths relative to /root/reference/Mila/Src/Dnn):
//   Compute/DeviceType.ixx:23-29            DeviceType (Rocm already exists in the enum)
//   Compute/IExecutionContext.ixx:26-62     IExecutionContext
//   Compute/ExecutionContext.ixx:24-117     ExecutionContext<TDeviceType>
//   Compute/Devices/Cuda/CudaExecutionContext.ixx:106-368   behaviour template for <Rocm>:
//       one stream, grow-only scratch that must be FETCHED PER FORWARD (never cached), synchronize()
//   Compute/ExecutionContextFactory.ixx:23-40   createExecutionContext (gains the Rocm case)
//   Tensors/TensorDataType.ixx:35-53, TensorDataTypeTraits.ixx:57-366 (PrecisionSupportedOnDevice
//       gains the Rocm clause), Tensors/Tensor.ixx:138-140, ITensor.ixx:39, Tensor.Types.ixx
//   Core/Component.BuildContext.ixx, Core/Model.RuntimeMode.ixx:23-27
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <numeric>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include <rocprofiler-sdk-roctx/roctx.h>

#include "mila_cdna4.h"

namespace Mila::Dnn
{
    // ---------------------------------------------------------------------------------------
    // Tracing: ROCTx ranges (the counterpart of the reference's NVTX ranges) around the host-side phases --
    // prefill, decode step, graph replay, sampler, checkpoint load.  `rocprofv3 --marker-trace` shows them on the timeline beside the
    // kernels; without a profiler attached a push / pop pair is a few tens of nanoseconds.
    // ---------------------------------------------------------------------------------------
    namespace Compute
    {
        class TraceRange
        {
        public:
            explicit TraceRange( const char* name ) noexcept { roctxRangePushA( name ); }
            ~TraceRange() { roctxRangePop(); }
            TraceRange( const TraceRange& ) = delete;
            TraceRange& operator=( const TraceRange& ) = delete;
        };
    }

    // ---------------------------------------------------------------------------------------
    // Errors: the C ABI's status codes become the reference's exception types
    // ---------------------------------------------------------------------------------------
    namespace Compute
    {
        /// counterpart of CudaException (Compute/Devices/Cuda/Helpers/CudaUtils.h)
        class RocmException : public std::runtime_error
        {
        public:
            explicit RocmException( const std::string& what ) : std::runtime_error( what ) {}
        };

        inline void rocmCheck( int status )
        {
            if ( status == MILA_OK ) return;
            const std::string text = mila_cdna4_last_error();
            switch ( status )
            {
                case MILA_E_INVALID_ARGUMENT: throw std::invalid_argument( text );
                case MILA_E_UNSUPPORTED: throw std::runtime_error( text );
                case MILA_E_SCRATCH_TOO_SMALL: throw std::runtime_error( text );
                default: throw RocmException( text );
            }
        }

        enum class DeviceType
        {
            Cpu,
            Cuda,
            Metal,
            Rocm,   ///< AMD CDNA4 (gfx950 / MI355X) through libmila_cdna4
        };

        inline std::string deviceTypeToString( DeviceType t )
        {
            switch ( t )
            {
                case DeviceType::Cpu: return "CPU";
                case DeviceType::Cuda: return "CUDA";
                case DeviceType::Metal: return "Metal";
                case DeviceType::Rocm: return "ROCm";
            }
            throw std::invalid_argument( "Invalid DeviceType" );
        }

        struct DeviceId
        {
            DeviceType type{ DeviceType::Cpu };
            int index{ 0 };
            bool operator==( const DeviceId& ) const = default;
        };

        struct Device
        {
            static DeviceId Cpu() { return { DeviceType::Cpu, 0 }; }
            static DeviceId Rocm( int index = 0 ) { return { DeviceType::Rocm, index }; }
        };
    }

    // ---------------------------------------------------------------------------------------
    // TensorDataType + traits
    // ---------------------------------------------------------------------------------------
    enum class TensorDataType
    {
        FP32, FP16, BF16, FP8_E4M3, FP8_E5M2, FP4_E2M1, FP4_E3M0, INT8, INT16, INT32, UINT8, UINT16, UINT32,
    };
    using dtype_t = TensorDataType;

    template<TensorDataType T> struct TensorDataTypeTraits;
    template<> struct TensorDataTypeTraits<TensorDataType::FP32> { using host_type = float; static constexpr size_t size_in_bytes = 4; static constexpr const char* name = "FP32"; };
    template<> struct TensorDataTypeTraits<TensorDataType::BF16> { using host_type = uint16_t; static constexpr size_t size_in_bytes = 2; static constexpr const char* name = "BF16"; };
    template<> struct TensorDataTypeTraits<TensorDataType::FP16> { using host_type = uint16_t; static constexpr size_t size_in_bytes = 2; static constexpr const char* name = "FP16"; };
    template<> struct TensorDataTypeTraits<TensorDataType::FP8_E4M3> { using host_type = uint8_t; static constexpr size_t size_in_bytes = 1; static constexpr const char* name = "FP8_E4M3"; };
    template<> struct TensorDataTypeTraits<TensorDataType::UINT8> { using host_type = uint8_t; static constexpr size_t size_in_bytes = 1; static constexpr const char* name = "UINT8"; };
    template<> struct TensorDataTypeTraits<TensorDataType::INT32> { using host_type = int32_t; static constexpr size_t size_in_bytes = 4; static constexpr const char* name = "INT32"; };

    /// Which compute precisions a device implements (TensorDataTypeTraits.ixx:361-366 + Rocm clause).
    template<TensorDataType TPrecision, Compute::DeviceType TDevice>
    concept PrecisionSupportedOnDevice =
        ( TDevice == Compute::DeviceType::Cpu && ( TPrecision == TensorDataType::FP32 || TPrecision == TensorDataType::INT32 ) ) ||
        ( TDevice == Compute::DeviceType::Rocm && ( TPrecision == TensorDataType::BF16 || TPrecision == TensorDataType::FP32 ||
                                                    TPrecision == TensorDataType::INT32 || TPrecision == TensorDataType::UINT8 ||
                                                    TPrecision == TensorDataType::FP8_E4M3 ) );

    using dim_t = int64_t;
    using shape_t = std::vector<dim_t>;

    inline dim_t shapeSize( const shape_t& s )
    {
        dim_t n = 1;
        for ( dim_t d : s ) n *= d;
        return n;
    }

    /// Tensors/Tensor.Types.ixx narrowToKernelIndex: shapes are narrowed to int once, at the launch site
    inline int narrowToKernelIndex( dim_t v, const char* what )
    {
        if ( v < 0 || v > 0x7fffffffLL ) throw std::overflow_error( std::string( what ) + " does not fit a kernel index" );
        return static_cast<int>( v );
    }

    inline std::string shapeToString( const shape_t& s )
    {
        std::string r = "[";
        for ( size_t i = 0; i < s.size(); ++i ) { if ( i ) r += ","; r += std::to_string( s[ i ] ); }
        return r + "]";
    }

    // ---------------------------------------------------------------------------------------
    // Memory resources
    // ---------------------------------------------------------------------------------------
    namespace Compute
    {
        struct CpuMemoryResource
        {
            static constexpr DeviceType device_type = DeviceType::Cpu;
            static void* allocate( size_t bytes ) { void* p = std::calloc( bytes ? bytes : 1, 1 ); if ( !p ) throw std::bad_alloc(); return p; }
            static void deallocate( void* p ) noexcept { std::free( p ); }
        };

        /// counterpart of CudaDeviceMemoryResource
        struct RocmDeviceMemoryResource
        {
            static constexpr DeviceType device_type = DeviceType::Rocm;
            static void* allocate( size_t bytes ) { void* p = nullptr; rocmCheck( mila_cdna4_malloc( &p, bytes ) ); return p; }
            static void deallocate( void* p ) noexcept { if ( p ) mila_cdna4_free( p ); }
        };

        template<DeviceType T> struct DeviceTypeTraits;
        template<> struct DeviceTypeTraits<DeviceType::Cpu> { using memory_resource = CpuMemoryResource; };
        template<> struct DeviceTypeTraits<DeviceType::Rocm> { using memory_resource = RocmDeviceMemoryResource; };
    }

    // ---------------------------------------------------------------------------------------
    // ITensor / Tensor
    // ---------------------------------------------------------------------------------------
    class ITensor
    {
    public:
        virtual ~ITensor() = default;
        virtual TensorDataType getDataType() const noexcept = 0;
        virtual const shape_t& shape() const noexcept = 0;
        virtual size_t size() const noexcept = 0;
        virtual void* rawData() noexcept = 0;
        virtual const void* rawData() const noexcept = 0;
        virtual Compute::DeviceType getDeviceType() const noexcept = 0;
        virtual const std::string& getName() const noexcept = 0;
    };

    template<TensorDataType TDataType, typename TMemoryResource>
    class Tensor : public ITensor
    {
    public:
        using host_type = typename TensorDataTypeTraits<TDataType>::host_type;
        static constexpr size_t kElemBytes = TensorDataTypeTraits<TDataType>::size_in_bytes;

        Tensor() = default;

        Tensor( Compute::DeviceId device, const shape_t& shape ) : device_( device ), shape_( shape )
        {
            if ( device.type != TMemoryResource::device_type )
                throw std::invalid_argument( "Tensor: device type does not match the memory resource" );
            for ( dim_t d : shape ) if ( d < 0 ) throw std::invalid_argument( "Tensor: negative dimension" );
            size_ = static_cast<size_t>( shapeSize( shape ) );
            void* p = TMemoryResource::allocate( size_ * kElemBytes );
            owner_ = std::shared_ptr<void>( p, []( void* q ) { TMemoryResource::deallocate( q ); } );
            data_ = p;
        }

        /// non-owning view over the same storage with a different shape (Tensor::view)
        Tensor view( const shape_t& new_shape ) const
        {
            if ( static_cast<size_t>( shapeSize( new_shape ) ) > size_ )
                throw std::invalid_argument( "Tensor::view: view " + shapeToString( new_shape ) + " exceeds storage " + shapeToString( shape_ ) );
            Tensor v;
            v.device_ = device_; v.shape_ = new_shape; v.size_ = static_cast<size_t>( shapeSize( new_shape ) );
            v.owner_ = owner_; v.data_ = data_; v.name_ = name_;
            return v;
        }

        /// view starting at an element offset (used for KV-prefix / last-token slices)
        Tensor slice( size_t element_offset, const shape_t& new_shape ) const
        {
            if ( element_offset + static_cast<size_t>( shapeSize( new_shape ) ) > size_ )
                throw std::out_of_range( "Tensor::slice: out of range" );
            Tensor v = view( new_shape );
            v.data_ = static_cast<unsigned char*>( data_ ) + element_offset * kElemBytes;
            return v;
        }

        TensorDataType getDataType() const noexcept override { return TDataType; }
        const shape_t& shape() const noexcept override { return shape_; }
        size_t size() const noexcept override { return size_; }
        void* rawData() noexcept override { return data_; }
        const void* rawData() const noexcept override { return data_; }
        host_type* data() noexcept { return static_cast<host_type*>( data_ ); }
        const host_type* data() const noexcept { return static_cast<const host_type*>( data_ ); }
        Compute::DeviceType getDeviceType() const noexcept override { return TMemoryResource::device_type; }
        Compute::DeviceId getDeviceId() const noexcept { return device_; }
        const std::string& getName() const noexcept override { return name_; }
        void setName( const std::string& n ) { name_ = n; }
        size_t sizeInBytes() const noexcept { return size_ * kElemBytes; }
        bool empty() const noexcept { return size_ == 0; }

    private:
        Compute::DeviceId device_{};
        shape_t shape_{};
        size_t size_{ 0 };
        std::shared_ptr<void> owner_{};
        void* data_{ nullptr };
        std::string name_{};
    };

    // ---------------------------------------------------------------------------------------
    // Execution contexts
    // ---------------------------------------------------------------------------------------
    namespace Compute
    {
        class IExecutionContext
        {
        public:
            virtual ~IExecutionContext() = default;
            [[nodiscard]] virtual DeviceId getDeviceId() const noexcept = 0;
            virtual void synchronize() = 0;
            [[nodiscard]] virtual std::size_t getScratchHighWaterBytes() const noexcept { return 0; }
        protected:
            IExecutionContext() = default;
        };

        template<DeviceType TDeviceType> class ExecutionContext;

        template<> class ExecutionContext<DeviceType::Cpu> : public IExecutionContext
        {
        public:
            explicit ExecutionContext( DeviceId id = Device::Cpu() ) : id_( id ) {}
            DeviceId getDeviceId() const noexcept override { return id_; }
            void synchronize() override {}
        private:
            DeviceId id_;
        };

        /// One HIP stream + grow-only device scratch.  Scratch pointers must be fetched on every
        /// forward and never cached by an op (reference rule, CudaLinearOp.ixx:603-614): a later,
        /// larger request re-allocates the block.
        template<> class ExecutionContext<DeviceType::Rocm> : public IExecutionContext
        {
        public:
            explicit ExecutionContext( DeviceId id = Device::Rocm( 0 ) ) : id_( id )
            {
                if ( id.type != DeviceType::Rocm ) throw std::invalid_argument( "ExecutionContext<Rocm>: device type mismatch" );
                int count = 0;
                rocmCheck( mila_cdna4_device_count( &count ) );
                if ( id.index < 0 || id.index >= count ) throw std::runtime_error( "ExecutionContext<Rocm>: no such device" );
                rocmCheck( mila_cdna4_set_device( id.index ) );
                rocmCheck( mila_cdna4_stream_create( &stream_ ) );
            }
            ~ExecutionContext() override
            {
                if ( scratch_ ) mila_cdna4_free( scratch_ );
                if ( stream_ ) mila_cdna4_stream_destroy( stream_ );
            }
            ExecutionContext( const ExecutionContext& ) = delete;
            ExecutionContext& operator=( const ExecutionContext& ) = delete;

            DeviceId getDeviceId() const noexcept override { return id_; }
            void synchronize() override { rocmCheck( mila_cdna4_stream_synchronize( stream_ ) ); }
            mila_stream_t getStream() const noexcept { return stream_; }
            /// adopt an externally owned stream (e.g. the host framework's current stream)
            void useExternalStream( mila_stream_t s ) { if ( stream_ && own_stream_ ) mila_cdna4_stream_destroy( stream_ ); stream_ = s; own_stream_ = false; }
            /// every op enqueues on getStream(): a caller that runs part of its work on a second stream swaps it in for the duration of those enqueues
            /// (RAII: StreamScope); ownership of the context's own stream is untouched
            mila_stream_t swapStream( mila_stream_t s ) noexcept { mila_stream_t old = stream_; stream_ = s; return old; }

            void* getScratch( size_t bytes )
            {
                if ( bytes > scratch_bytes_ )
                {
                    synchronize();
                    if ( scratch_ ) rocmCheck( mila_cdna4_free( scratch_ ) );
                    scratch_ = nullptr;
                    rocmCheck( mila_cdna4_malloc( &scratch_, bytes ) );
                    scratch_bytes_ = bytes;
                }
                return scratch_;
            }
            std::size_t getScratchHighWaterBytes() const noexcept override { return scratch_bytes_; }

        private:
            DeviceId id_;
            mila_stream_t stream_{ nullptr };
            bool own_stream_{ true };
            void* scratch_{ nullptr };
            size_t scratch_bytes_{ 0 };
        };

        using RocmExecutionContext = ExecutionContext<DeviceType::Rocm>;
        using CpuExecutionContext = ExecutionContext<DeviceType::Cpu>;

        /// Compute/ExecutionContextFactory.ixx:23-40 with the Rocm case the reference lacks
        inline std::unique_ptr<IExecutionContext> createExecutionContext( DeviceId id )
        {
            switch ( id.type )
            {
                case DeviceType::Cpu: return std::make_unique<CpuExecutionContext>( id );
                case DeviceType::Rocm: return std::make_unique<RocmExecutionContext>( id );
                default: throw std::runtime_error( "createExecutionContext: unsupported device type " + deviceTypeToString( id.type ) );
            }
        }

        template<DeviceType T>
        inline ExecutionContext<T>* cast_context( IExecutionContext* ctx )
        {
            if ( !ctx || ctx->getDeviceId().type != T ) throw std::invalid_argument( "cast_context: execution context is null or of the wrong device type" );
            return static_cast<ExecutionContext<T>*>( ctx );
        }
    }

    // ---------------------------------------------------------------------------------------
    // Host <-> device transfer (Compute/Devices/Cuda/Tensors: copy/convert subset the path needs)
    // ---------------------------------------------------------------------------------------
    template<TensorDataType T>
    inline void copyToDevice( Tensor<T, Compute::RocmDeviceMemoryResource>& dst, const void* host_src, size_t bytes,
                              Compute::RocmExecutionContext* ctx )
    {
        if ( bytes > dst.sizeInBytes() ) throw std::invalid_argument( "copyToDevice: source larger than destination" );
        Compute::rocmCheck( mila_cdna4_memcpy_h2d( dst.rawData(), host_src, bytes, ctx->getStream() ) );
        ctx->synchronize();
    }

    template<TensorDataType T>
    inline void copyToHost( void* host_dst, const Tensor<T, Compute::RocmDeviceMemoryResource>& src, size_t bytes,
                            Compute::RocmExecutionContext* ctx )
    {
        if ( bytes > src.sizeInBytes() ) throw std::invalid_argument( "copyToHost: destination larger than source" );
        Compute::rocmCheck( mila_cdna4_memcpy_d2h( host_dst, src.rawData(), bytes, ctx->getStream() ) );
        ctx->synchronize();
    }

    // ---------------------------------------------------------------------------------------
    // MemoryStats (Core/Component.MemoryStats.ixx:64-160): the footprint contract every component carries -- getMemoryStats() = what it holds now,
    // getRequiredMemory(BuildContext) = what build() (and, for quantized Linears, the load that follows) would allocate, without allocating.  The two must
    // agree after a real build; tests/test_memory_stats_gpu.py is the drift gate (Tests/Dnn/Components/Transformers/Gemma/Gemma.Cuda.cpp:207-239).
    // ---------------------------------------------------------------------------------------
    struct MemoryStats
    {
        std::size_t device_parameter_bytes{ 0 };      ///< weights, scales, biases
        std::size_t device_state_bytes{ 0 };          ///< output buffers, KV caches, RoPE tables, op-owned derived state (resident prefill staging)
        std::size_t device_gradient_bytes{ 0 };       ///< always 0: the CDNA4 backend is inference-only
        std::size_t host_parameter_bytes{ 0 }, host_state_bytes{ 0 }, host_gradient_bytes{ 0 };
        [[nodiscard]] std::size_t totalDeviceBytes() const noexcept { return device_parameter_bytes + device_state_bytes + device_gradient_bytes; }
        [[nodiscard]] std::size_t totalHostBytes() const noexcept { return host_parameter_bytes + host_state_bytes + host_gradient_bytes; }
        [[nodiscard]] std::size_t totalBytes() const noexcept { return totalDeviceBytes() + totalHostBytes(); }
        MemoryStats& operator+=( const MemoryStats& r ) noexcept
        {
            device_parameter_bytes += r.device_parameter_bytes; device_state_bytes += r.device_state_bytes; device_gradient_bytes += r.device_gradient_bytes;
            host_parameter_bytes += r.host_parameter_bytes; host_state_bytes += r.host_state_bytes; host_gradient_bytes += r.host_gradient_bytes;
            return *this;
        }
        bool operator==( const MemoryStats& ) const = default;
    };
    /// bytes a tensor handle holds (0 for an empty handle)
    template<typename TPtr> inline std::size_t tensorBytes( const TPtr& t ) { return t ? t->sizeInBytes() : 0; }

    // ---------------------------------------------------------------------------------------
    // BuildContext
    // ---------------------------------------------------------------------------------------
    enum class RuntimeMode : uint8_t { Inference, Training };

    class BuildContext
    {
    public:
        BuildContext() = default;
        BuildContext( const shape_t& input_shape, RuntimeMode mode, bool output_installed = false, dim_t prefill_chunk = 0 )
            : input_shape_( input_shape ), mode_( mode ), output_installed_( output_installed ), prefill_chunk_( prefill_chunk )
        {
            if ( input_shape.empty() ) throw std::invalid_argument( "BuildContext: input_shape is empty" );
        }
        const shape_t& inputShape() const noexcept { return input_shape_; }
        RuntimeMode runtimeMode() const noexcept { return mode_; }
        bool isOutputInstalled() const noexcept { return output_installed_; }
        dim_t prefillChunkSize() const noexcept { return prefill_chunk_; }
    private:
        shape_t input_shape_{ 1 };
        RuntimeMode mode_{ RuntimeMode::Inference };
        bool output_installed_{ false };
        dim_t prefill_chunk_{ 0 };
    };
}
