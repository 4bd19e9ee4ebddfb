// Tahoe/Math/Array.h -- minimal growable host array (reference: Tahoe/Math/Array.h:22-98; only the
// members the sort path and its test touch: sized ctor, operator[], begin, getSize, setSize, pushBack).
#pragma once
#include <Tahoe/Math/Math.h>
#include <string.h>

namespace Tahoe {

template <typename T>
class Array {
public:
    Array() : m_data(0), m_size(0), m_capacity(0) {}
    explicit Array(u64 size) : m_data(0), m_size(0), m_capacity(0) { setSize(size); }
    ~Array() { delete[] m_data; }

    T& operator[](u64 i) { return m_data[i]; }
    const T& operator[](u64 i) const { return m_data[i]; }
    T* begin() { return m_data; }
    const T* begin() const { return m_data; }
    T* end() { return m_data + m_size; }
    u64 getSize() const { return m_size; }
    void clear() { m_size = 0; }

    void setSize(u64 size)
    {
        if (size > m_capacity) {
            u64 cap = m_capacity ? m_capacity : 16;
            while (cap < size) cap *= 2;
            T* grown = new T[cap];
            for (u64 i = 0; i < m_size; ++i) grown[i] = m_data[i];
            delete[] m_data;
            m_data = grown;
            m_capacity = cap;
        }
        m_size = size;
    }
    void pushBack(const T& v)
    {
        setSize(m_size + 1);
        m_data[m_size - 1] = v;
    }

private:
    Array(const Array&);
    Array& operator=(const Array&);
    T* m_data;
    u64 m_size, m_capacity;
};

}  // namespace Tahoe
