// Stand-in for the three MFEM containers parelagmc_amd/host/mfem_adapter.hpp touches, with exactly the members it uses
// (test infrastructure: lets CI compile and run the adapter on a machine without MFEM; never shipped).
#pragma once
#include <vector>

namespace mfem {

class Vector {
  public:
    Vector() = default;
    explicit Vector(int n) : d_((size_t)n, 0.0) {}
    Vector(const double* p, int n) : d_(p, p + n) {}
    int Size() const { return (int)d_.size(); }
    void SetSize(int n) { d_.resize((size_t)n); }
    double* GetData() const { return const_cast<double*>(d_.data()); }
    double& operator()(int i) { return d_[(size_t)i]; }
    const double& operator()(int i) const { return d_[(size_t)i]; }

  private:
    std::vector<double> d_;
};

template <class T>
class Array {
  public:
    Array() = default;
    Array(const T* p, int n) : d_(p, p + n) {}
    int Size() const { return (int)d_.size(); }
    T* GetData() const { return const_cast<T*>(d_.data()); }
    T& operator[](int i) { return d_[(size_t)i]; }
    const T& operator[](int i) const { return d_[(size_t)i]; }

  private:
    std::vector<T> d_;
};

class SparseMatrix {
  public:
    SparseMatrix() = default;
    SparseMatrix(int h, int w, const int* I, const int* J, const double* A) : h_(h), w_(w), I_(I, I + h + 1), J_(J, J + I[h]), A_(A, A + I[h]) {}
    int Height() const { return h_; }
    int Width() const { return w_; }
    const int* GetI() const { return I_.data(); }
    const int* GetJ() const { return J_.data(); }
    const double* GetData() const { return A_.data(); }

  private:
    int h_ = 0, w_ = 0;
    std::vector<int> I_, J_;
    std::vector<double> A_;
};

}  // namespace mfem
