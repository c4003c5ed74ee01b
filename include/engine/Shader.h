// engine/Shader.h -- API shell of toyraygun::Shader (reference src/engine/Shader.h:61-73).
// The reference loads MSL/HLSL text at run time and compiles it through the backend; the HIP kernels
// are compiled ahead of time, so a Shader here is a registry of names: load() records the path,
// addFunction() the entry points the app names (main.cpp:27-31,46,61-62), compile() succeeds.
#pragma once
#include <sstream>
#include <string>
#include <vector>

#include "engine/Engine.h"

namespace toyraygun {

enum class ShaderType { None = 0, Graphics, Compute, Raytrace, Count };

enum class ShaderFunctionType {
    None = 0,
    Vertex, Fragment,                                   // graphics
    Compute,                                            // compute
    RayGen, ClosestHit, Miss, ShadowHit, ShadowMiss,    // raytracing
    Count
};

struct ShaderFunction {
    std::string functionName;
    ShaderFunctionType functionType;
};

class Shader {
public:
    static std::vector<std::string> s_skipShaderIncludes;

    std::string m_path;
    std::string m_sourcePath;
    std::stringstream m_sourceText;

    virtual ~Shader() {}
    virtual bool load(std::string path, bool preprocess = true);
    virtual void preprocess();
    virtual bool compile(ShaderType type);

    virtual void addFunction(std::string functionName, ShaderFunctionType functionType = ShaderFunctionType::None);
    virtual std::vector<std::string> getFunctionNames();
    virtual std::string getFunction(ShaderFunctionType functionType);
    virtual std::wstring getFunctionW(ShaderFunctionType functionType);

    virtual std::string getSourceText();
    virtual void *getBufferPointer(ShaderFunctionType type = ShaderFunctionType::None);
    virtual size_t getBufferSize(ShaderFunctionType type = ShaderFunctionType::None);
    virtual void *getCompiledShader(ShaderFunctionType type = ShaderFunctionType::None);

protected:
    std::vector<ShaderFunction> m_functions;
    ShaderType m_compiledAs = ShaderType::None;
};

}  // namespace toyraygun
