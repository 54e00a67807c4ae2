// Component layer (L3): Component<Dev,Prec> lifecycle and the leaf components of the forward() hot
// path.  Mirrors /root/reference/Mila/Src/Dnn/Core/Component.ixx:155-1100 (lifecycle
// setExecutionContext -> build(BuildContext) -> loadParameter(name, blob) -> forward(...)) and
// Components/Linear/Linear.ixx:82-1093 -- the Linear<Device, Precision, TWeightQuant> template surface
// is kept verbatim in meaning; with Device = DeviceType::Rocm it resolves, through OperationTraits,
// to the CDNA4 op classes of Operations.h.
#pragma once

#include <map>

#include "Operations.h"

namespace Mila::Dnn
{
    using Compute::DeviceId;
    using Compute::DeviceType;
    using Compute::IExecutionContext;
    using Compute::OperationTraits;
    using Compute::OperationType;
    using Quant::Weight::NoWeightQuant;
    using Quant::Weight::WeightQuantPolicy;

    // ---------------------------------------------------------------------------------------
    // Component base
    // ---------------------------------------------------------------------------------------
    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class Component
    {
    public:
        explicit Component( const std::string& name ) : name_( name )
        {
            if ( name.empty() ) throw std::invalid_argument( "Component: name must not be empty" );
        }
        virtual ~Component() = default;

        static constexpr DeviceType getDeviceType() { return TDeviceType; }
        static constexpr TensorDataType getPrecision() { return TPrecision; }
        const std::string& getName() const noexcept { return name_; }
        bool isBuilt() const noexcept { return built_; }

        void setExecutionContext( IExecutionContext* ctx )
        {
            if ( !ctx ) throw std::invalid_argument( name_ + ": execution context is null" );
            if ( ctx->getDeviceId().type != TDeviceType ) throw std::invalid_argument( name_ + ": execution context device type mismatch" );
            if ( context_ ) throw std::runtime_error( name_ + ": execution context already set" );
            context_ = ctx;
            onExecutionContextSet();
        }
        IExecutionContext* getExecutionContext() const
        {
            if ( !context_ ) throw std::runtime_error( name_ + ": no execution context" );
            return context_;
        }

        void build( const BuildContext& ctx )
        {
            if ( built_ ) throw std::runtime_error( name_ + ": build() called twice" );
            if ( !context_ ) throw std::runtime_error( name_ + ": setExecutionContext() must precede build()" );
            onBuilding( ctx );
            built_ = true;
        }

        /// what this component holds on the device right now (Component.ixx getMemoryStats)
        virtual MemoryStats getMemoryStats() const { return {}; }
        /// what build( ctx ) would allocate, without allocating (Component.ixx getRequiredMemory); equals getMemoryStats() after that build
        virtual MemoryStats getRequiredMemory( const BuildContext& ) const { return {}; }

        /// host blob in the checkpoint's layout; throws std::invalid_argument for unknown names / sizes
        virtual void loadParameter( const std::string& param_name, const void* host_blob, size_t bytes )
        {
            (void)host_blob; (void)bytes;
            throw std::invalid_argument( name_ + ": unknown parameter '" + param_name + "'" );
        }

    protected:
        virtual void onExecutionContextSet() {}
        virtual void onBuilding( const BuildContext& ) = 0;

        std::string name_;
        IExecutionContext* context_{ nullptr };
        bool built_{ false };
    };

    // ---------------------------------------------------------------------------------------
    // Linear
    // ---------------------------------------------------------------------------------------
    class LinearConfig
    {
    public:
        LinearConfig( dim_t in_features, dim_t out_features ) : in_( in_features ), out_( out_features ) {}
        template<typename Self> decltype( auto ) withBias( this Self&& self, bool b ) { self.bias_ = b; return std::forward<Self>( self ); }
        dim_t getInputFeatures() const noexcept { return in_; }
        dim_t getOutputFeatures() const noexcept { return out_; }
        bool hasBias() const noexcept { return bias_; }
        void validate() const
        {
            if ( in_ <= 0 || out_ <= 0 ) throw std::invalid_argument( "LinearConfig: Input and output features must be greater than zero" );
        }
    private:
        dim_t in_, out_;
        bool bias_{ true };
    };

    template<DeviceType TDeviceType, TensorDataType TComputePrecision, WeightQuantPolicy TWeightQuant = NoWeightQuant>
        requires PrecisionSupportedOnDevice<TComputePrecision, TDeviceType>
    class Linear : public Component<TDeviceType, TComputePrecision>
    {
    public:
        using ComponentBase = Component<TDeviceType, TComputePrecision>;
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TComputePrecision, MR>;
        using OpType = typename OperationTraits<OperationType::LinearOp, TDeviceType, TComputePrecision, TWeightQuant>::type;

        static constexpr bool kIsQuantized = TWeightQuant::kIsQuantized;
        static constexpr TensorDataType kWeightDtype = kIsQuantized ? TWeightQuant::kStorageDtype : TComputePrecision;
        using WeightTensorType = Tensor<kWeightDtype, MR>;
        using WeightScaleTensorType = Tensor<TWeightQuant::kScaleDtype, MR>;

        explicit Linear( const std::string& name, const LinearConfig& config, std::optional<DeviceId> device_id = std::nullopt )
            : ComponentBase( name ), config_( config )
        {
            config_.validate();
            if ( device_id.has_value() )
            {
                if ( device_id->type != TDeviceType ) throw std::invalid_argument( "Linear: device type mismatch" );
                owned_exec_context_ = Compute::createExecutionContext( device_id.value() );
                this->setExecutionContext( owned_exec_context_.get() );
            }
        }

        /// output = input * weight^T + bias; returns the component-owned output or a shape-adjusted view
        TensorType& forward( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Linear must be built before calling forward." );
            validateInputShape( input.shape() );
            auto out_shape = input.shape();
            out_shape.back() = config_.getOutputFeatures();
            if ( input.shape() == leading_shape_ && !output_installed_ )
            {
                operation_->forward( input, *output_ );
                return *output_;
            }
            output_view_ = std::make_unique<TensorType>( output_->view( out_shape ) );
            operation_->forward( input, *output_view_ );
            return *output_view_;
        }

        /// true when forwardGelu() serves an input of this shape (unquantized weights, more than one row)
        bool fusesGelu( const shape_t& in_shape ) const noexcept
        {
            dim_t rows = 1;
            for ( size_t i = 0; i + 1 < in_shape.size(); ++i ) rows *= in_shape[ i ];
            return this->isBuilt() && operation_->fusesGelu( rows );
        }

        /// forward() followed by a tanh-GELU on its output, in one kernel: same bits, the intermediate stays in registers
        TensorType& forwardGelu( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Linear must be built before calling forward." );
            validateInputShape( input.shape() );
            auto out_shape = input.shape();
            out_shape.back() = config_.getOutputFeatures();
            if ( input.shape() == leading_shape_ && !output_installed_ )
            {
                operation_->forwardGelu( input, *output_ );
                return *output_;
            }
            output_view_ = std::make_unique<TensorType>( output_->view( out_shape ) );
            operation_->forwardGelu( input, *output_view_ );
            return *output_view_;
        }

        /// forward() on rows the producer already quantized per token (W4A8 policy; see RocmLinearOp::acceptsFp8Activations): `leading` is the shape of the
        /// bf16 input those rows stand for; returns the component-owned output or a shape-adjusted view, with the bits forward() would have produced
        TensorType& forwardFp8Activations( const uint8_t* x8, const float* ts, const shape_t& leading )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Linear must be built before calling forward." );
            validateInputShape( leading );
            auto out_shape = leading;
            out_shape.back() = config_.getOutputFeatures();
            dim_t rows = 1;
            for ( size_t i = 0; i + 1 < leading.size(); ++i ) rows *= leading[ i ];
            output_view_ = std::make_unique<TensorType>( output_->view( out_shape ) );
            operation_->forwardFp8Activations( x8, ts, output_view_->data(), static_cast<int>( rows ) );
            return *output_view_;
        }

        TensorType& backward( const TensorType&, const TensorType& )
        {
            if constexpr ( kIsQuantized ) throw std::logic_error( "Linear: backward is not supported on quantized weights" );
            throw std::runtime_error( "Linear: built for inference; backward requires RuntimeMode::Training" );
        }

        /// "weight": bf16 [N,K] blob -> stored as is, or quantized on load (Linear.ixx:529-558);
        /// already-packed blob of the storage dtype + "weight_scale" -> direct copies (:559-574)
        void loadParameter( const std::string& param_name, const void* host_blob, size_t bytes ) override
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Linear: build() must precede loadParameter()" );
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            const size_t N = static_cast<size_t>( config_.getOutputFeatures() ), K = static_cast<size_t>( config_.getInputFeatures() );
            constexpr size_t EB = TensorType::kElemBytes;      // bytes per element of an unquantized blob: 2 (bf16), or 4 on the FP32 row
            if ( param_name == "weight" )
            {
                if ( bytes == N * K * EB )
                {
                    if constexpr ( kIsQuantized )
                    {
                        void* staging = ctx->getScratch( bytes );
                        Compute::rocmCheck( mila_cdna4_memcpy_h2d( staging, host_blob, bytes, ctx->getStream() ) );
                        operation_->quantize( static_cast<const uint16_t*>( staging ), *weight_, *weight_scale_ );
                        ctx->synchronize();
                        operation_->onQuantizedWeightsLoaded();
                    }
                    else
                        copyToDevice( *weight_, host_blob, bytes, ctx );
                }
                else if ( kIsQuantized && bytes == weight_->sizeInBytes() )
                    copyToDevice( *weight_, host_blob, bytes, ctx );
                else
                    throw std::invalid_argument( this->getName() + ": weight blob has " + std::to_string( bytes ) + " bytes, expected " + std::to_string( N * K * EB ) );
            }
            else if ( param_name == "weight_scale" )
            {
                if constexpr ( kIsQuantized )
                {
                    if ( bytes != weight_scale_->sizeInBytes() ) throw std::invalid_argument( this->getName() + ": weight_scale blob size mismatch" );
                    copyToDevice( *weight_scale_, host_blob, bytes, ctx );
                    operation_->onQuantizedWeightsLoaded();
                }
                else throw std::invalid_argument( this->getName() + ": unquantized Linear has no weight_scale" );
            }
            else if ( param_name == "bias" )
            {
                if ( !bias_ ) throw std::invalid_argument( this->getName() + ": configured without bias" );
                if ( bytes != N * EB ) throw std::invalid_argument( this->getName() + ": bias blob size mismatch" );
                copyToDevice( *bias_, host_blob, bytes, ctx );
            }
            else
                throw std::invalid_argument( this->getName() + ": unknown parameter '" + param_name + "'" );
        }

        /// quantize-on-load from a bf16 [N,K] tensor already resident on the device
        void loadWeightFromDevice( const uint16_t* device_bf16 )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "Linear: build() must precede loadWeightFromDevice()" );
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            if constexpr ( kIsQuantized ) { operation_->quantize( device_bf16, *weight_, *weight_scale_ ); operation_->onQuantizedWeightsLoaded(); }
            else Compute::rocmCheck( mila_cdna4_memcpy_d2d( weight_->rawData(), device_bf16, weight_->sizeInBytes(), ctx->getStream() ) );
        }

        /// adopt another component's table instead of allocating (the tied lm_head: Linear.ixx:614-680, Gemma.ixx:659-684); precedes build()
        /// Tying contract (Linear.ixx:614-680; Tests/.../Linear.Cuda.cpp:618-641): an unquantized Linear adopts a weight; a per-channel FP8 one adopts (weight, scales) --
        /// the per-output-channel scale axis IS the vocab row a tied embedding gathers; a quantized weight WITHOUT scales, and every per-group policy (input-axis scales
        /// do not transfer to a row gather), are rejected with std::logic_error before anything is dereferenced
        void installSharedWeight( std::shared_ptr<WeightTensorType> shared_weight )
        {
            if constexpr ( kIsQuantized ) throw std::logic_error( this->getName() + ": a quantized Linear cannot adopt a weight without its scales" );
            else
            {
                checkShared( shared_weight.get() );
                weight_ = std::move( shared_weight );
            }
        }
        void installSharedWeight( std::shared_ptr<WeightTensorType> shared_weight, std::shared_ptr<WeightScaleTensorType> shared_scales )
        {
            if constexpr ( !kIsQuantized ) throw std::logic_error( this->getName() + ": an unquantized Linear has no weight scales to adopt" );
            else if constexpr ( !TWeightQuant::kPerChannel ) throw std::logic_error( this->getName() + ": per-group scales run along the input axis and cannot be tied to a row gather" );
            else
            {
            checkShared( shared_weight.get() );
            if ( !shared_scales || shared_scales->size() != static_cast<size_t>( config_.getOutputFeatures() ) )
                throw std::invalid_argument( this->getName() + ": installSharedWeight needs one scale per output feature" );
            weight_ = std::move( shared_weight );
            weight_scale_ = std::move( shared_scales );
            }
        }
        bool hasSharedWeight() const noexcept { return shared_weight_; }
        /// a model whose layers run one after the other hands every layer's Linear of one role the same output buffer (the reference's pooled block workspace,
        /// Gemma.ixx allocateBlockWorkspace / installSharedOutput); precedes build(), the owner accounts for the buffer
        void installSharedOutput( std::shared_ptr<TensorType> output )
        {
            if ( this->isBuilt() ) throw std::runtime_error( this->getName() + ": installSharedOutput() must precede build()" );
            output_ = std::move( output );
            output_installed_ = true;
        }

        /// Linear.ixx:692-833.  An installed (tied) weight IS reported -- the tying composite subtracts it exactly once; an installed output is not (its owner counts it).
        /// State: the output buffer and the op-owned derived state (fp4 tensor scale, resident prefill staging -- present once quantized weights are in place).
        MemoryStats getMemoryStats() const override
        {
            MemoryStats st;
            st.device_parameter_bytes = tensorBytes( weight_ ) + tensorBytes( weight_scale_ ) + tensorBytes( bias_ );
            if ( !output_installed_ ) st.device_state_bytes += tensorBytes( output_ );
            if ( operation_ ) st.device_state_bytes += operation_->stateBytes();
            return st;
        }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            validateInputShape( ctx.inputShape() );
            MemoryStats st;
            const size_t N = static_cast<size_t>( config_.getOutputFeatures() ), K = static_cast<size_t>( config_.getInputFeatures() );
            if ( shared_weight_ ) st.device_parameter_bytes += tensorBytes( weight_ ) + tensorBytes( weight_scale_ );
            else if constexpr ( !kIsQuantized ) st.device_parameter_bytes += N * K * WeightTensorType::kElemBytes;
            else if constexpr ( TWeightQuant::kPerChannel ) st.device_parameter_bytes += N * K * WeightTensorType::kElemBytes + N * WeightScaleTensorType::kElemBytes;
            else st.device_parameter_bytes += N * ( K / 2 ) * WeightTensorType::kElemBytes + N * ( K / Quant::Weight::groupSizeOf<TWeightQuant>() ) * WeightScaleTensorType::kElemBytes;
            if ( config_.hasBias() ) st.device_parameter_bytes += N * TensorType::kElemBytes;
            if ( !output_installed_ && !ctx.isOutputInstalled() ) st.device_state_bytes += static_cast<size_t>( shapeSize( ctx.inputShape() ) / ctx.inputShape().back() ) * N * TensorType::kElemBytes;
            if ( operation_ ) st.device_state_bytes += operation_->requiredStateBytes();
            return st;
        }

        WeightTensorType& getWeight() { return *weight_; }
        WeightScaleTensorType* getWeightScale() { return weight_scale_.get(); }
        TensorType* getBias() { return bias_.get(); }
        const LinearConfig& getConfig() const noexcept { return config_; }
        OpType& getOperation() { return *operation_; }

        /// parameter bytes resident on the device (Linear.ixx:692-833 MemoryStats, parameters only)
        size_t getParameterBytes() const
        {
            size_t b = weight_ ? weight_->sizeInBytes() : 0;
            if ( weight_scale_ ) b += weight_scale_->sizeInBytes();
            if ( bias_ ) b += bias_->sizeInBytes();
            return b;
        }

    protected:
        void onExecutionContextSet() override
        {
            operation_ = std::make_shared<OpType>( this->getExecutionContext(),
                                                   Compute::LinearOpConfig{ config_.getInputFeatures(), config_.getOutputFeatures(), config_.hasBias() } );
        }

        void onBuilding( const BuildContext& ctx ) override
        {
            validateInputShape( ctx.inputShape() );
            const auto dev = this->getExecutionContext()->getDeviceId();
            const dim_t N = config_.getOutputFeatures(), K = config_.getInputFeatures();
            // initializeParameters (Linear.ixx:1029-1054); an installed shared weight is kept
            if ( shared_weight_ ) {}
            else if constexpr ( !kIsQuantized ) weight_ = std::make_shared<WeightTensorType>( dev, shape_t{ N, K } );
            else if constexpr ( TWeightQuant::kPerChannel )
            {
                weight_ = std::make_shared<WeightTensorType>( dev, shape_t{ N, K } );
                weight_scale_ = std::make_shared<WeightScaleTensorType>( dev, shape_t{ N } );
            }
            else
            {
                constexpr int G = Quant::Weight::groupSizeOf<TWeightQuant>();
                if ( K % G != 0 ) throw std::invalid_argument( this->getName() + ": in_features must be a multiple of the quantization group size" );
                weight_ = std::make_shared<WeightTensorType>( dev, shape_t{ N, K / 2 } );
                weight_scale_ = std::make_shared<WeightScaleTensorType>( dev, shape_t{ N, K / G } );
            }
            if ( !shared_weight_ ) weight_->setName( this->getName() + ".weight" );
            if ( config_.hasBias() ) bias_ = std::make_shared<TensorType>( dev, shape_t{ N } );
            operation_->setParameters( weight_.get(), bias_.get() );
            if constexpr ( kIsQuantized ) operation_->setWeightScales( weight_scale_.get() );
            operation_->build( ctx );
            leading_shape_ = ctx.inputShape();
            auto out_shape = leading_shape_;
            out_shape.back() = N;
            if ( !output_installed_ ) output_ = std::make_shared<TensorType>( dev, out_shape );
            else if ( output_->size() < static_cast<size_t>( shapeSize( out_shape ) ) ) throw std::invalid_argument( this->getName() + ": the installed output is too small" );
        }

    private:
        void checkShared( const WeightTensorType* w )
        {
            if ( this->isBuilt() ) throw std::runtime_error( this->getName() + ": installSharedWeight() must precede build()" );
            const size_t N = static_cast<size_t>( config_.getOutputFeatures() ), K = static_cast<size_t>( config_.getInputFeatures() );
            if ( !w || w->size() != N * K ) throw std::invalid_argument( this->getName() + ": the shared weight must hold [out_features, in_features] elements" );
            shared_weight_ = true;
        }
        void validateInputShape( const shape_t& s ) const
        {
            if ( s.empty() || s.back() != config_.getInputFeatures() )
                throw std::invalid_argument( this->getName() + ": input feature dimension " + ( s.empty() ? std::string( "<none>" ) : std::to_string( s.back() ) ) +
                                             " does not match in_features " + std::to_string( config_.getInputFeatures() ) );
            if ( !leading_shape_.empty() && shapeSize( s ) / s.back() > shapeSize( leading_shape_ ) / leading_shape_.back() )
                throw std::invalid_argument( this->getName() + ": input " + shapeToString( s ) + " exceeds the built shape " + shapeToString( leading_shape_ ) );
        }

        LinearConfig config_;
        std::unique_ptr<IExecutionContext> owned_exec_context_;
        std::shared_ptr<OpType> operation_;
        std::shared_ptr<WeightTensorType> weight_;
        std::shared_ptr<WeightScaleTensorType> weight_scale_;
        std::shared_ptr<TensorType> bias_;
        std::shared_ptr<TensorType> output_;
        std::unique_ptr<TensorType> output_view_;
        shape_t leading_shape_;
        bool shared_weight_{ false }, output_installed_{ false };
    };

    // ---------------------------------------------------------------------------------------
    // RmsNorm (Components/Normalization/RmsNorm/RmsNorm.ixx; config defaults RmsNorm.Config.ixx:289-293)
    // ---------------------------------------------------------------------------------------
    class RmsNormConfig
    {
    public:
        explicit RmsNormConfig( dim_t dim ) : dim_( dim ) {}
        template<typename Self> decltype( auto ) withEpsilon( this Self&& s, float e ) { s.eps_ = e; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withBias( this Self&& s, bool b ) { s.bias_ = b; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withUnitOffset( this Self&& s, bool u ) { s.unit_offset_ = u; return std::forward<Self>( s ); }
        template<typename Self> decltype( auto ) withWeight( this Self&& s, bool w ) { s.weight_ = w; return std::forward<Self>( s ); }
        dim_t dim() const noexcept { return dim_; }
        float epsilon() const noexcept { return eps_; }
        bool hasBias() const noexcept { return bias_; }
        bool hasWeight() const noexcept { return weight_; }
        bool unitOffset() const noexcept { return unit_offset_; }
        void validate() const
        {
            if ( dim_ <= 0 ) throw std::invalid_argument( "RmsNormConfig: normalized dimension must be positive" );
            if ( !( eps_ > 0.0f ) ) throw std::invalid_argument( "RmsNormConfig: epsilon must be positive" );
        }
    private:
        dim_t dim_;
        float eps_{ 1e-5f };
        bool bias_{ true }, weight_{ true }, unit_offset_{ false };
    };

    template<DeviceType TDeviceType, TensorDataType TPrecision>
    class RmsNorm : public Component<TDeviceType, TPrecision>
    {
    public:
        using MR = typename Compute::DeviceTypeTraits<TDeviceType>::memory_resource;
        using TensorType = Tensor<TPrecision, MR>;
        using OpType = typename OperationTraits<OperationType::RmsNormOp, TDeviceType, TPrecision>::type;

        RmsNorm( const std::string& name, const RmsNormConfig& cfg ) : Component<TDeviceType, TPrecision>( name ), config_( cfg ) { config_.validate(); }

        TensorType& forward( const TensorType& input )
        {
            if ( !this->isBuilt() ) throw std::runtime_error( "RmsNorm must be built before calling forward." );
            if ( input.shape().back() != config_.dim() ) throw std::invalid_argument( this->getName() + ": trailing dimension mismatch" );
            if ( input.size() > output_->size() ) throw std::invalid_argument( this->getName() + ": input exceeds the built shape" );
            view_ = std::make_unique<TensorType>( output_->view( input.shape() ) );
            operation_->forward( input, *view_ );
            return *view_;
        }
        /// in place over a caller tensor (per-head q/k/v norms on views, Gemma.Block.ixx:228-233)
        void forwardInto( const TensorType& input, TensorType& output ) { operation_->forward( input, output ); }

        void loadParameter( const std::string& n, const void* blob, size_t bytes ) override
        {
            auto* ctx = Compute::cast_context<TDeviceType>( this->getExecutionContext() );
            if ( n == "weight" && weight_ ) { if ( bytes != weight_->sizeInBytes() ) throw std::invalid_argument( this->getName() + ": weight blob size mismatch" ); copyToDevice( *weight_, blob, bytes, ctx ); }
            else if ( n == "bias" && bias_ ) { if ( bytes != bias_->sizeInBytes() ) throw std::invalid_argument( this->getName() + ": bias blob size mismatch" ); copyToDevice( *bias_, blob, bytes, ctx ); }
            else throw std::invalid_argument( this->getName() + ": unknown parameter '" + n + "'" );
        }
        TensorType* getWeight() { return weight_.get(); }
        const RmsNormConfig& getConfig() const noexcept { return config_; }
        OpType& getOperation() { return *operation_; }
        /// see Linear::installSharedOutput
        void installSharedOutput( std::shared_ptr<TensorType> output )
        {
            if ( this->isBuilt() ) throw std::runtime_error( this->getName() + ": installSharedOutput() must precede build()" );
            output_ = std::move( output );
            output_installed_ = true;
        }
        /// RmsNorm.ixx:329-420: parameters weight (+ bias); state the output buffer (unless installed) + the op's per-slice rstd
        MemoryStats getMemoryStats() const override
        {
            MemoryStats st;
            st.device_parameter_bytes = tensorBytes( weight_ ) + tensorBytes( bias_ );
            if ( !output_installed_ ) st.device_state_bytes += tensorBytes( output_ );
            if ( operation_ ) st.device_state_bytes += operation_->stateBytes();
            return st;
        }
        MemoryStats getRequiredMemory( const BuildContext& ctx ) const override
        {
            MemoryStats st;
            const size_t D = static_cast<size_t>( config_.dim() ), n = static_cast<size_t>( shapeSize( ctx.inputShape() ) );
            st.device_parameter_bytes = ( config_.hasWeight() ? D : 0 ) * TensorType::kElemBytes + ( config_.hasBias() ? D : 0 ) * TensorType::kElemBytes;
            if ( !output_installed_ && !ctx.isOutputInstalled() ) st.device_state_bytes += n * TensorType::kElemBytes;
            st.device_state_bytes += ( n / D ) * TensorType::kElemBytes;      // rstd, one per slice, in the compute dtype
            return st;
        }

    protected:
        void onExecutionContextSet() override
        {
            operation_ = std::make_shared<OpType>( this->getExecutionContext(),
                Compute::NormOpConfig{ config_.dim(), config_.epsilon(), config_.hasBias(), config_.unitOffset() ? 1.0f : 0.0f } );
        }
        void onBuilding( const BuildContext& ctx ) override
        {
            const auto dev = this->getExecutionContext()->getDeviceId();
            if ( config_.hasWeight() ) weight_ = std::make_shared<TensorType>( dev, shape_t{ config_.dim() } );
            if ( config_.hasBias() ) bias_ = std::make_shared<TensorType>( dev, shape_t{ config_.dim() } );
            operation_->setParameters( weight_.get(), bias_.get() );
            operation_->build( ctx );
            if ( !output_installed_ ) output_ = std::make_shared<TensorType>( dev, ctx.inputShape() );
            else if ( output_->size() < static_cast<size_t>( shapeSize( ctx.inputShape() ) ) ) throw std::invalid_argument( this->getName() + ": the installed output is too small" );
        }
    private:
        RmsNormConfig config_;
        std::shared_ptr<OpType> operation_;
        std::shared_ptr<TensorType> weight_, bias_, output_;
        std::unique_ptr<TensorType> view_;
        bool output_installed_{ false };
    };

    // ---------------------------------------------------------------------------------------
    // Small helpers used by the block wiring
    // ---------------------------------------------------------------------------------------
    /// uploads a bf16 parameter generated on the host
    template<typename TTensor>
    inline void uploadBf16( TTensor& t, const std::vector<uint16_t>& host, Compute::RocmExecutionContext* ctx )
    {
        if ( host.size() * 2 != t.sizeInBytes() ) throw std::invalid_argument( "uploadBf16: size mismatch" );
        copyToDevice( t, host.data(), host.size() * 2, ctx );
    }
}
