// Shader.cpp -- name registry standing in for run-time shader compilation (include/engine/Shader.h).
#include "engine/Shader.h"

namespace toyraygun {

std::vector<std::string> Shader::s_skipShaderIncludes;

// The three shader names the reference app loads (main.cpp:24,41,56) map onto kernels that are already
// inside libtoyraygun_hip.so; any other name is "not found", like a missing file in the reference.
bool Shader::load(std::string path, bool doPreprocess) {
    m_path = path;
    m_sourcePath = Engine::getRuntimeShaderPath() + path + "." + Engine::getRuntimeShaderExt();
    m_sourceText.str(std::string());
    m_sourceText.clear();
    const bool known = path == "Raytracing" || path == "Accumulate" || path == "PostProcessing";
    if (!known) return false;
    m_sourceText << "// " << path << ": compiled ahead of time into libtoyraygun_hip.so (gfx950)\n";
    if (doPreprocess) preprocess();
    return true;
}
void Shader::preprocess() {}
bool Shader::compile(ShaderType type) {
    m_compiledAs = type;
    return type != ShaderType::None && type != ShaderType::Count;
}
void Shader::addFunction(std::string functionName, ShaderFunctionType functionType) {
    ShaderFunction f;
    f.functionName = functionName;
    f.functionType = functionType;
    m_functions.push_back(f);
}
std::vector<std::string> Shader::getFunctionNames() {
    std::vector<std::string> names;
    for (size_t i = 0; i < m_functions.size(); ++i) names.push_back(m_functions[i].functionName);
    return names;
}
std::string Shader::getFunction(ShaderFunctionType functionType) {
    for (size_t i = 0; i < m_functions.size(); ++i)
        if (m_functions[i].functionType == functionType) return m_functions[i].functionName;
    return "";
}
std::wstring Shader::getFunctionW(ShaderFunctionType functionType) {
    const std::string s = getFunction(functionType);
    return std::wstring(s.begin(), s.end());
}
std::string Shader::getSourceText() { return m_sourceText.str(); }
void *Shader::getBufferPointer(ShaderFunctionType) { return nullptr; }
size_t Shader::getBufferSize(ShaderFunctionType) { return 0; }
void *Shader::getCompiledShader(ShaderFunctionType) { return nullptr; }

}  // namespace toyraygun
