#ifndef GLX_ORACLE_STUB_DAG_PB_H_
#define GLX_ORACLE_STUB_DAG_PB_H_
namespace graphlearn {
class DagDef {};
class DagNodeDef {};
class DagEdgeDef {};
}  // namespace graphlearn
#endif
